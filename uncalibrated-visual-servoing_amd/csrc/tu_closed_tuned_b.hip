#define UVS_TU_SHAPES UVS_TUNED_SHAPES_B
#define UVS_TU_NAME closed_tuned_b
#include "tu_closed_tuned.inc"
