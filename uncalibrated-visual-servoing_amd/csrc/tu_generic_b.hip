#define UVS_TU_SHAPES UVS_SHAPES_B
#define UVS_TU_CLOSED closed_generic_b
#define UVS_TU_REPLAY replay_generic_b
#include "tu_generic.inc"
