// camera-model translation unit: mix14 (ND=14, projection family decided per camera at run time) -- rigs that mix pinhole
// and fisheye cameras; every camera is padded to the widest coefficient block -- see mcba_cam_impl.h
#define MCBA_ND 14
#define MCBA_FISH 2
#define MCBA_CAM_FN cam_ops_mix14
#include "mcba_cam_impl.h"
