"""gpuacceleratedtracking_amd -- MI355X-native GNSS tracking correlator.

Host-side mirror (Python) of the operator surface the reference (coezmaden/
GPUAcceleratedTracking + Tracking.jl) exposes for ONE path: downconvert + correlate.  All
compute runs in libgat.so (hand-written HIP for gfx950) through the C ABI of include/gat.h.
Importing this package never touches the test oracle and there is no CPU fallback: without the
HIP library / a HIP device the operators raise.
"""
from ._lib import GAT_FLAG_GRAPH, GAT_MC_AUTO, GAT_MC_BF16_SPLIT, GAT_MC_F32, GAT_MC_VECTOR  # noqa: F401
from ._lib import (GAT_FLAG_ATOMIC, GAT_LAYOUT_INTERLEAVED, GAT_LAYOUT_INTERLEAVED_I8,  # noqa: F401
                   GAT_LAYOUT_INTERLEAVED_I16, GAT_LAYOUT_PLANAR, SAMPLE_BYTES, GatError,
                   library_path, load as load_library)
from .algorithms import (ALGODICT, ALGODICTINV, MEMDICT, REDDICT, KernelAlgorithm, ReductionAlgorithm,  # noqa: F401
                         ReplicaAlgorithm, cpu_reduce_partial_sum, cuda_reduce_partial_sum, kernel_algorithm)
from .benchmarks import (add_metadata, add_results, algorithmic_bytes, build_stream,  # noqa: F401
                         run_kernel_benchmark, run_reduction_benchmark, run_replica_benchmark, stream_scenario)
from .context import Context, ResidentCorrelator, get_context  # noqa: F401
from .correlator import (EarlyPromptLateCorrelator, NumAccumulators, NumAnts, get_accumulators,  # noqa: F401
                         get_correlator_sample_shifts, get_num_accumulators, get_num_ants)
from .gen_signal import StructSignal, gen_blank_signal, gen_signal, gen_signal_stream, make_params  # noqa: F401
from .loop import ResidentTrackingLoop, TrackingLoop  # noqa: F401
from .sharding import DeviceGroup, ShardPlan, gather_outputs, shard_channels, shard_params  # noqa: F401
from .signals import GNSSDICT, GPSL1, GPSL5, generate_codes, get_code_frequency, get_code_length  # noqa: F401
from .tracking import (StreamCorrelator, downconvert_and_accumulate_strided, downconvert_and_correlate,  # noqa: F401
                       gen_code_replica, gen_code_replica_nsat, reduce_cplx_multi)

__version__ = "0.1.0"
