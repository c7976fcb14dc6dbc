// stand-in for <rccl/rccl.h> (types only; api.hip dlopen's the library and the stubbed build never asks for the RCCL gather)
#pragma once
#include <stddef.h>
typedef struct h2stub_comm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
