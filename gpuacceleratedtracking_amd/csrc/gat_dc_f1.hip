// gat_dc_f1.hip -- instances of the fused correlator kernel (gat_dc.h) for sample format GAT_LAYOUT_INTERLEAVED.
// One translation unit per format so that the ~150 instances of each compile in parallel.
#include "gat_dc.h"

namespace gat {
template hipError_t launch_dc_fmt<GAT_LAYOUT_INTERLEAVED>(const DcArgs &, const DcLaunch &, hipStream_t);
} // namespace gat
