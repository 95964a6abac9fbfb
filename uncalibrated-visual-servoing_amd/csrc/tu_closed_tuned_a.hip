#define UVS_TU_SHAPES UVS_TUNED_SHAPES_A
#define UVS_TU_NAME closed_tuned_a
#include "tu_closed_tuned.inc"
