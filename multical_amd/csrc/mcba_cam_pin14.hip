// camera-model translation unit: pin14 (ND=14, fisheye=false) -- see mcba_cam_impl.h
#define MCBA_ND 14
#define MCBA_FISH 0
#define MCBA_CAM_FN cam_ops_pin14
#include "mcba_cam_impl.h"
