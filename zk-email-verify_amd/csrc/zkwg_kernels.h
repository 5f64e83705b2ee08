// Kernel declarations (definitions in zkwg_kernels_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "zkwg_sched.h"

__global__ void zk_sha_chain(ZkSched s, const u8* in, u32* hst, u32 n_emails);
__global__ void zk_sha_expand(ZkSched s, const u8* in, const u32* hst, uint4* wit, u32 n_emails);
__global__ void zk_misc_sha_main(ZkSched s, const u8* in, const u32* hst, const uint4* invtab,
                                 uint4* wit_all, int* status, u32 n_emails);
#define ZK_EXPAND_WAVES 4
