#define UVS_TU_CAREFUL_SHAPES UVS_CAREFUL_SHAPES_B
#define UVS_TU_CLOSED_CAREFUL closed_careful_b
#define UVS_TU_REPLAY_CAREFUL replay_careful_b
#include "tu_careful.inc"
