// gat_resident_f2.hip -- instances of the resident correlator (gat_resident.h) for sample format GAT_LAYOUT_INTERLEAVED_I16.
#include "gat_resident.h"

namespace gat {
template hipError_t launch_dc_resident_fmt<GAT_LAYOUT_INTERLEAVED_I16>(const DcArgs &, const DcLaunch &, const ResidentArgs &, hipStream_t, int *);
} // namespace gat
