// Host runtime of the framework (no GPU code): checkpoint bundle I/O, data
// pipeline, parameter-server service.  Plain C ABI for ctypes.
#pragma once
#include <stddef.h>
#include <stdint.h>

#define DTF_RT_EXPORT extern "C" __attribute__((visibility("default")))

DTF_RT_EXPORT int dtf_runtime_abi_version();
