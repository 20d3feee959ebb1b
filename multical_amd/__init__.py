"""multical_amd -- MI355X-native bundle-adjustment back-end for multical (one hot path, HIP only).

    from multical_amd import Calibration, Workspace      # host mirror of the reference classes
    from multical_amd import dropin; dropin.install()    # patch a real multical installation

See DESIGN.md (path, layout, kernels) and INTEGRATION.md (how multical binds include/mcba.h).
"""
from .structs import struct, Table                                   # noqa: F401
from .parameters import ParamList                                    # noqa: F401
from .pose_set import PoseSet                                        # noqa: F401
from .motion import StaticFrames, RollingFrames, HandEye             # noqa: F401
from .camera import Camera, CameraFisheye                            # noqa: F401
from .board import Board, CharucoBoard, AprilGrid                    # noqa: F401
from .calibration import Calibration, select_threshold, error_stats  # noqa: F401
from .workspace import Workspace                                     # noqa: F401
