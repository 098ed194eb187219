"""Host mirror of the LocalRegNeuralDE.jl operator surface for the adaptive
Tsit5 neural-ODE path, over the liblrnde C ABI (include/lrnde.h).

Mirrors (reference paths): NeuralODE — src/layers/neural_ode.jl; TDChain —
src/layers/common.jl:2-45; diffeqsol_to_array / diffeqsol_to_timeseries —
src/utils.jl:37-46.  PyTorch is used only for device memory and streams.
"""
from ._lib import LIB_PATH, LrndeError, set_option  # noqa: F401  (raises if liblrnde.so is missing)
from .layers import (Chain, Dense, Handle, NeuralODE, ODESolution, TDChain,  # noqa: F401
                     diffeqsol_to_array, diffeqsol_to_timeseries, flatten_params,
                     glorot_params)
from .sde import NeuralDSDE, SdeHandle  # noqa: F401
from .conv import BatchNorm, Conv, ConvHandle, glorot_conv_params  # noqa: F401
from .training import run_cifar_training_step, run_training_step  # noqa: F401
from .optim import (Constant, CosineAnneal, ExponentialDecay, InverseDecay, Optimiser, Step,  # noqa: F401
                    construct_scheduler)
from .sharding import LocalComm, init_comm, run_ranks, shard_columns  # noqa: F401
