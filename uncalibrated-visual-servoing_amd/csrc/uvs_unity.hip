// Single-translation-unit view of the library for the diagnostic builds (make quick / stamps / ablate / asm), which pass one set of
// -D flags to everything.  The shipped library is built from the separate translation units (make, make -j).
#include "uvs_rmckf.hip"
#include "tu_closed_tuned_a.hip"
#undef UVS_TU_SHAPES
#undef UVS_TU_NAME
#include "tu_closed_tuned_b.hip"
#undef UVS_TU_SHAPES
#undef UVS_TU_NAME
#include "tu_generic_a.hip"
#undef UVS_TU_SHAPES
#undef UVS_TU_CLOSED
#undef UVS_TU_REPLAY
#include "tu_generic_b.hip"
#include "tu_careful_a.hip"
#undef UVS_TU_CAREFUL_SHAPES
#undef UVS_TU_CLOSED_CAREFUL
#undef UVS_TU_REPLAY_CAREFUL
#include "tu_careful_b.hip"
#include "tu_closed_wide.hip"
#include "tu_replay_tuned.hip"
#include "tu_replay_f32.hip"
#include "tu_misc.hip"
