// camera-model translation unit: pin5 (ND=5, fisheye=false) -- see mcba_cam_impl.h
#define MCBA_ND 5
#define MCBA_FISH 0
#define MCBA_CAM_FN cam_ops_pin5
#include "mcba_cam_impl.h"
