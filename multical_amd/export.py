"""Calibration -> JSON in multical's export format (multical/io/export_calib.py:9-97, FORMAT.md), so that results of the
HIP back-end open in stock multical tooling (`multical vis`, downstream rig loaders).

Only needed by the standalone mirror classes: with `multical_amd.dropin.install()` the objects ARE the reference's
`Calibration` and its own exporter runs unchanged.
"""
import json

import numpy as np


def export_camera(camera):
  """export_calib.py:9-15"""
  return dict(model=camera.model, image_size=_plain(camera.image_size), K=np.asarray(camera.intrinsic).tolist(),
              dist=np.asarray(camera.dist).tolist())


def export_cameras(camera_names, cameras):
  """export_calib.py:17-18"""
  return {k: export_camera(camera) for k, camera in zip(camera_names, cameras)}


def export_transform(pose):
  """export_calib.py:20-22: R = pose[:3, :3], T = pose[:3, 3] (transform/matrix.py:26-30)."""
  pose = np.asarray(pose)
  return dict(R=pose[:3, :3].tolist(), T=pose[:3, 3].tolist())


def export_camera_poses(camera_names, camera_poses):
  """export_calib.py:25-28: valid poses only."""
  return {k: export_transform(pose)
          for k, pose, valid in zip(camera_names, camera_poses.poses, camera_poses.valid) if valid}


def export_relative(camera_names, camera_poses, master):
  """export_calib.py:31-36: keys `<camera>_to_<master>`, the master under its own name."""
  assert master in camera_names
  return {k if master == k else f"{k}_to_{master}": export_transform(pose)
          for k, pose, valid in zip(camera_names, camera_poses.poses, camera_poses.valid) if valid}


def export_images(camera_names, filenames):
  """export_calib.py:59-63"""
  return dict(rgb=[{camera: image for image, camera in zip(images, camera_names)} for images in filenames])


def export_json(calib, names, filenames, master=None):
  """export_calib.py:81-97.  `names.camera`: camera names; `filenames`: per camera, the list of image names."""
  if master is not None:
    calib = calib.with_master(master)
  camera_names = list(names.camera if hasattr(names, "camera") else names["camera"])
  camera_poses = calib.camera_poses.pose_table
  filenames = [list(x) for x in zip(*filenames)] if len(filenames) else []   # structs.struct.transpose_lists
  return dict(
    cameras=export_cameras(camera_names, calib.cameras),
    camera_poses=export_camera_poses(camera_names, camera_poses) if master is None
    else export_relative(camera_names, camera_poses, master),
    image_sets=export_images(camera_names, filenames))


def export(filename, calib, names, filenames, master=None):
  """export_calib.py:74-78"""
  data = export_json(calib, names, filenames, master=master)
  with open(filename, "w") as outfile:
    json.dump(data, outfile, indent=2)


def _plain(x):
  return [int(v) for v in x] if isinstance(x, (tuple, list, np.ndarray)) else x
