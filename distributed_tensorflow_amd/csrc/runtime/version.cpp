#include "dtf_runtime.h"

DTF_RT_EXPORT int dtf_runtime_abi_version() { return 1; }
