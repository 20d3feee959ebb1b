// camera-model translation unit: pin4 (ND=4, fisheye=false) -- see mcba_cam_impl.h
#define MCBA_ND 4
#define MCBA_FISH 0
#define MCBA_CAM_FN cam_ops_pin4
#include "mcba_cam_impl.h"
