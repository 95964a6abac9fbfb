#define UVS_TU_SHAPES UVS_SHAPES_A
#define UVS_TU_CLOSED closed_generic_a
#define UVS_TU_REPLAY replay_generic_a
#include "tu_generic.inc"
