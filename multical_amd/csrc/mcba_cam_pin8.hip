// camera-model translation unit: pin8 (ND=8, fisheye=false) -- see mcba_cam_impl.h
#define MCBA_ND 8
#define MCBA_FISH 0
#define MCBA_CAM_FN cam_ops_pin8
#include "mcba_cam_impl.h"
