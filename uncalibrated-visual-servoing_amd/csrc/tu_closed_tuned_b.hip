#include "launchers.hpp"
#define UVS_TU_SHAPES UVS_TUNED_SHAPES_B
#define UVS_TU_NAME closed_tuned_b
#ifdef UVS_HAVE_EMU2
#define UVS_TU_EMU2
#endif
#include "tu_closed_tuned.inc"
