// gat_dc_f0.hip -- instances of the fused correlator kernel (gat_dc.h) for sample format GAT_LAYOUT_PLANAR.
// One translation unit per format so that the ~150 instances of each compile in parallel.
#include "gat_dc.h"

namespace gat {
template hipError_t launch_dc_fmt<GAT_LAYOUT_PLANAR>(const DcArgs &, const DcLaunch &, hipStream_t);
// the instance table for the host's planner (here: rebuilt together with the kernels by development variants)
bool dc_has_instance(int ant_tile, int taps, int vec, int aw, int kt, int nw, int depth) { return dc_instance(ant_tile, taps, vec, aw, kt, nw, depth); }
} // namespace gat
