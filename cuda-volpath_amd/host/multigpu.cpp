// multigpu.cpp -- see multigpu.h
#include "multigpu.h"

#include <dlfcn.h>

#include <set>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

namespace volpath
{
namespace
{
// the RCCL entry points used, resolved at run time
struct Rccl
{
    decltype(&ncclCommInitAll)    CommInitAll    = nullptr;
    decltype(&ncclCommDestroy)    CommDestroy    = nullptr;
    decltype(&ncclReduce)         Reduce         = nullptr;
    decltype(&ncclGroupStart)     GroupStart     = nullptr;
    decltype(&ncclGroupEnd)       GroupEnd       = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
} R;

template <class F>
bool sym(void* lib, const char* name, F& f)
{
    f = reinterpret_cast<F>(dlsym(lib, name));
    return f != nullptr;
}
}  // namespace

bool NodeReducer::init(const std::vector<int>& devices, std::string& err)
{
    devices_ = devices;
    std::set<int> distinct(devices.begin(), devices.end());
    if (devices.size() < 2 || distinct.size() != devices.size()) return true;  // one GPU, or contexts sharing a GPU: no collective
    lib_ = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib_) lib_ = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib_) { err = std::string("cannot load librccl.so.1: ") + dlerror(); return false; }
    if (!sym(lib_, "ncclCommInitAll", R.CommInitAll) || !sym(lib_, "ncclCommDestroy", R.CommDestroy) || !sym(lib_, "ncclReduce", R.Reduce) ||
        !sym(lib_, "ncclGroupStart", R.GroupStart) || !sym(lib_, "ncclGroupEnd", R.GroupEnd) || !sym(lib_, "ncclGetErrorString", R.GetErrorString))
    {
        err = "librccl.so.1 lacks an expected symbol";
        return false;
    }
    std::vector<ncclComm_t> comms(devices.size());
    ncclResult_t rc = R.CommInitAll(comms.data(), (int)devices.size(), devices.data());
    if (rc != ncclSuccess) { err = std::string("ncclCommInitAll: ") + R.GetErrorString(rc); return false; }
    for (auto c : comms) comms_.push_back((void*)c);
    return true;
}

bool NodeReducer::reduce_to_root(const std::vector<vp_ctx*>& ctx, const std::vector<vp_float4*>& acc, const std::vector<void*>& stream,
                                 size_t n_float4, std::string& err)
{
    const size_t n = acc.size();
    if (n < 2) return true;
    if (!comms_.empty())
    {
        // one reduce of W*H*4 floats to rank 0, all ranks' calls in one group (single host thread)
        ncclResult_t rc = R.GroupStart();
        for (size_t i = 0; i < n && rc == ncclSuccess; i++)
        {
            if (hipSetDevice(devices_[i]) != hipSuccess) { err = "hipSetDevice failed"; R.GroupEnd(); return false; }
            rc = R.Reduce(acc[i], acc[i], n_float4 * 4, ncclFloat, ncclSum, 0, (ncclComm_t)comms_[i], (hipStream_t)stream[i]);
        }
        ncclResult_t rc2 = R.GroupEnd();
        if (rc == ncclSuccess) rc = rc2;
        if (rc != ncclSuccess) { err = std::string("ncclReduce: ") + R.GetErrorString(rc); return false; }
        return true;
    }
    // contexts share one GPU: their accumulators are in the same memory; add them on the root context's stream after each
    // context has finished rendering
    for (size_t i = 1; i < n; i++)
    {
        vp_ctx_set_current(ctx[i]);
        if (vp_synchronize()) { err = vp_last_error(); return false; }
    }
    vp_ctx_set_current(ctx[0]);
    for (size_t i = 1; i < n; i++)
        if (vp_accumulate(acc[0], acc[i], n_float4)) { err = vp_last_error(); return false; }
    return true;
}

NodeReducer::~NodeReducer()
{
    for (void* c : comms_) R.CommDestroy((ncclComm_t)c);
    if (lib_) dlclose(lib_);
}
}  // namespace volpath
