// recogym_hip.hip — librecogym_hip.so as ONE translation unit: the nine units of the library, in order (the parallel build
// compiles them separately: __graft_entry__.build()).  gfx950 only.
#include "rg_host.hip"
#include "rg_exact.hip"
#include "rg_draw_fp32.hip"
#include "rg_draw_pipelined.hip"
#include "rg_draw_wide.hip"
#include "rg_advance.hip"
#include "rg_walk.hip"
#include "rg_draw_exacthi.hip"
#include "rg_draw_lds.hip"
