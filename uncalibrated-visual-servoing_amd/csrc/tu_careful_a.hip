#define UVS_TU_CAREFUL_SHAPES UVS_CAREFUL_SHAPES_A
#define UVS_TU_CLOSED_CAREFUL closed_careful_a
#define UVS_TU_REPLAY_CAREFUL replay_careful_a
#include "tu_careful.inc"
