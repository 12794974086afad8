// smh_consts.h -- every tunable of the vision hot path, shared by host C++ and HIP device code.
// Analogue of the reference's toml-consts step (vision-common/build.rs:1-14 generates consts.rs and
// consts.cu from vision-common/src/consts/consts.toml so CPU and GPU share thresholds).
// Values: consts.toml:1-63; bounds: vision-common/src/consts/mod.rs:7-19.
#pragma once
#include <stdint.h>

#define SMH_BUTTON_R 217
#define SMH_BUTTON_G 67
#define SMH_BUTTON_B 49
#define SMH_BUTTON_TOLERANCE 25
#define SMH_BUTTON_RED_PIXEL_THRESHOLD 0.65f

#define SMH_OCR_BRIGHTNESS_THRESHOLD 200
#define SMH_OCR_MONOCHROMATICY_THRESHOLD 3
#define SMH_OCR_BRIGHTNESS_EDGE_THRESHOLD 130
#define SMH_OCR_SIMILARITY_EDGE_THRESHOLD 48
#define SMH_OCR_DILATE_RADIUS 3

#define SMH_ALPHA_H 105
#define SMH_ALPHA_S 100
#define SMH_ALPHA_V 100
#define SMH_BRAVO_H 285
#define SMH_BRAVO_S 46
#define SMH_BRAVO_V 85
#define SMH_CHARLIE_H 158
#define SMH_CHARLIE_S 60
#define SMH_CHARLIE_V 91
#define SMH_HSV_HUE_TOLERANCE 15
#define SMH_HSV_SAT_TOLERANCE 15
#define SMH_HSV_VIB_TOLERANCE 15
#define SMH_HSV_MIN_SAT 35
#define SMH_PLAYER_DIR_ARC_SAT 50

// image 0.23.14 sRGB luma weights (mirrored at vision-gpu/cuda/cuda.cu:23-25)
#define SMH_LUMA_R 0.2126f
#define SMH_LUMA_G 0.7152f
#define SMH_LUMA_B 0.0722f

// lsd (vision-common/src/lsd.rs:9,86,94; vision-cpu/src/lib.rs:434-437; vision-common/src/lib.rs:58)
#define SMH_LSD_RAYS 3600
#define SMH_LSD_ACCEPT_LEN_SQ 2500.0f
#define SMH_LSD_PROXIMITY_SQ 50.0f
#define SMH_LSD_CENTRE_REACH 5.0f
#define SMH_LSD_MAX_LINES 32
// a ray whose final gap starts at step K ends K-1 unit steps (+-0.25) from its start: K <= 49 => len^2 < 2500
#define SMH_LSD_REJECT_K 49u

// mpx_ratio.rs:5-6,12
#define SMH_MIN_SCALE_WIDTH 10u
#define SMH_MIN_SCALE_VERTICAL_BAR_HEIGHT 4u

// screen-relative bounds, all fractions of the screen HEIGHT (consts/mod.rs:7-19)
#define SMH_MAP_X 0.018522135
#define SMH_MAP_Y_BOTTOM 0.07421875
#define SMH_MAP_W 0.864930556
#define SMH_MAP_H 0.761078559
#define SMH_BTN_X_RIGHT 0.0078125
#define SMH_BTN_Y_BOTTOM 0.0078125
#define SMH_BTN_W 0.236132813
#define SMH_BTN_H 0.038205295
