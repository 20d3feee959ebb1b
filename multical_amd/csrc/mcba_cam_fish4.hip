// camera-model translation unit: fish4 (ND=4, fisheye=true) -- see mcba_cam_impl.h
#define MCBA_ND 4
#define MCBA_FISH 1
#define MCBA_CAM_FN cam_ops_fish4
#include "mcba_cam_impl.h"
