// camera-model translation unit: pin12 (ND=12, fisheye=false) -- see mcba_cam_impl.h
#define MCBA_ND 12
#define MCBA_FISH 0
#define MCBA_CAM_FN cam_ops_pin12
#include "mcba_cam_impl.h"
