// multigpu.h -- the single-process multi-GPU host (north_star: "pixel tiles shard across the 8 GPUs of one node with a
// final RCCL reduce of per-tile HDR accumulators over xGMI").  The reference has no counterpart: it is single-GPU
// (cudaSetDevice(0), src/denoiser.cpp:94-97).  One libvolpath_hip context per GPU renders its share of the 8x8 pixel tiles
// (vp_set_shard) into a full-frame accumulator that stays zero elsewhere; NodeReducer sums the accumulators onto GPU 0 with
// ONE ncclReduce(sum, float, W*H*4, root 0).  Disjoint tiles make that sum exact, so the N-GPU image is the 1-GPU image
// bit for bit.
#pragma once
#include <cstddef>
#include <string>
#include <vector>

#include "volpath.h"

namespace volpath
{
// The RCCL call sequence of NodeReducer on ONE device: load librccl.so.1, ncclCommInitAll for a single rank, one in-place
// ncclReduce(sum) of `count` floats on a stream of that device, compare, destroy.  It cannot show that ranks talk to each other --
// a one-GPU box has nobody to talk to -- but it does run every RCCL entry point this host uses on the hardware.
// (`volpath_render --rccl-selftest [device]`)
bool rccl_selftest(int device, size_t count, std::string& report);

class NodeReducer
{
public:
    // devices[i] = HIP device of rank i.  With all-distinct devices RCCL is loaded (librccl.so.1, dlopen: the 1-GPU driver
    // never needs it) and one communicator per rank is created by ncclCommInitAll.  With repeated devices (several contexts
    // on one GPU -- how the N > 1 path is exercised on a one-GPU box) there is nothing to communicate over and the sum is
    // vp_accumulate on the root context.
    bool init(const std::vector<int>& devices, std::string& err);
    // acc[i]: device pointer of rank i's accumulator (n float4), ctx[i] its context, stream[i] the hipStream_t the context
    // launches on (as void*).  On return the sum is in acc[0], complete on stream[0] order (callers synchronise ctx 0).
    bool reduce_to_root(const std::vector<vp_ctx*>& ctx, const std::vector<vp_float4*>& acc, const std::vector<void*>& stream,
                        size_t n_float4, std::string& err);
    bool uses_rccl() const { return !comms_.empty(); }
    // what the first run on more than one GPU should say about itself: the RCCL version, the number of communicators, and every
    // communicator's ncclCommCount / ncclCommUserRank (this path has never exchanged a byte between ranks)
    std::string describe() const;
    // destroys the communicators; call it BEFORE the contexts (whose streams the collectives ran on) are destroyed
    void shutdown();
    ~NodeReducer();

private:
    std::vector<int>   devices_;
    std::vector<void*> comms_;  // ncclComm_t per rank
    void*              lib_ = nullptr;
};
}  // namespace volpath
