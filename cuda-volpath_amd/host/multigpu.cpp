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
    decltype(&ncclGetVersion)     GetVersion     = nullptr;   // optional: only for describe()
    decltype(&ncclCommCount)      CommCount      = nullptr;
    decltype(&ncclCommUserRank)   CommUserRank   = nullptr;
} R;

template <class F>
bool sym(void* lib, const char* name, F& f)
{
    f = reinterpret_cast<F>(dlsym(lib, name));
    return f != nullptr;
}
}  // namespace

static bool load_rccl(void*& lib, std::string& err)
{
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { err = std::string("cannot load librccl.so.1: ") + dlerror(); return false; }
    if (!sym(lib, "ncclCommInitAll", R.CommInitAll) || !sym(lib, "ncclCommDestroy", R.CommDestroy) || !sym(lib, "ncclReduce", R.Reduce) ||
        !sym(lib, "ncclGroupStart", R.GroupStart) || !sym(lib, "ncclGroupEnd", R.GroupEnd) || !sym(lib, "ncclGetErrorString", R.GetErrorString))
    {
        err = "librccl.so.1 lacks an expected symbol";
        return false;
    }
    (void)sym(lib, "ncclGetVersion", R.GetVersion);
    (void)sym(lib, "ncclCommCount", R.CommCount);
    (void)sym(lib, "ncclCommUserRank", R.CommUserRank);
    return true;
}

bool rccl_selftest(int device, size_t count, std::string& report)
{
    void* lib = nullptr;
    if (!load_rccl(lib, report)) return false;
    bool        ok = false;
    ncclComm_t  comm = nullptr;
    float*      d = nullptr;
    hipStream_t st = nullptr;
    std::vector<float> h(count), back(count);
    for (size_t i = 0; i < count; i++) h[i] = (float)(i % 1000) * 0.25f - 7.0f;
    do
    {
        if (hipSetDevice(device) != hipSuccess) { report = "hipSetDevice failed"; break; }
        ncclResult_t rc = R.CommInitAll(&comm, 1, &device);
        if (rc != ncclSuccess) { report = std::string("ncclCommInitAll: ") + R.GetErrorString(rc); comm = nullptr; break; }
        if (hipStreamCreate(&st) != hipSuccess || hipMalloc((void**)&d, count * sizeof(float)) != hipSuccess) { report = "hipMalloc / hipStreamCreate failed"; break; }
        if (hipMemcpyAsync(d, h.data(), count * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess) { report = "upload failed"; break; }
        rc = R.GroupStart();
        if (rc == ncclSuccess) rc = R.Reduce(d, d, count, ncclFloat, ncclSum, 0, comm, st);
        ncclResult_t rc2 = R.GroupEnd();
        if (rc == ncclSuccess) rc = rc2;
        if (rc != ncclSuccess) { report = std::string("ncclReduce: ") + R.GetErrorString(rc); break; }
        if (hipMemcpyAsync(back.data(), d, count * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        {
            report = "download failed";
            break;
        }
        ok = back == h;   // the sum over one rank is the rank's own data
        int ver = 0, cnt = -1;
        if (R.GetVersion) (void)R.GetVersion(&ver);
        if (R.CommCount) (void)R.CommCount(comm, &cnt);
        report = ok ? "RCCL " + std::to_string(ver) + ", ncclCommCount " + std::to_string(cnt) + ": ncclCommInitAll(1 rank) + ncclReduce(sum, float, " +
                          std::to_string(count) + ") on device " + std::to_string(device) + ": ok"
                    : "ncclReduce over one rank changed the data";
    } while (false);
    if (d) (void)hipFree(d);
    if (st) (void)hipStreamDestroy(st);
    if (comm) R.CommDestroy(comm);
    dlclose(lib);
    return ok;
}

bool NodeReducer::init(const std::vector<int>& devices, std::string& err)
{
    devices_ = devices;
    std::set<int> distinct(devices.begin(), devices.end());
    if (devices.size() < 2 || distinct.size() != devices.size()) return true;  // one GPU, or contexts sharing a GPU: no collective
    if (!load_rccl(lib_, err)) return false;
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

std::string NodeReducer::describe() const
{
    if (comms_.empty()) return "no collective (one GPU, or contexts sharing a GPU: on-device sum)";
    int ver = 0;
    if (R.GetVersion) (void)R.GetVersion(&ver);
    std::string s = "RCCL " + std::to_string(ver) + ", " + std::to_string(comms_.size()) + " communicators (ncclCommInitAll), devices";
    for (int d : devices_) s += " " + std::to_string(d);
    s += "; ncclCommCount/ncclCommUserRank per communicator:";
    for (void* c : comms_)
    {
        int cnt = -1, rk = -1;
        if (R.CommCount) (void)R.CommCount((ncclComm_t)c, &cnt);
        if (R.CommUserRank) (void)R.CommUserRank((ncclComm_t)c, &rk);
        s += " " + std::to_string(cnt) + "/" + std::to_string(rk);
    }
    return s;
}

void NodeReducer::shutdown()
{
    for (void* c : comms_) R.CommDestroy((ncclComm_t)c);
    comms_.clear();
}
NodeReducer::~NodeReducer()
{
    shutdown();
    if (lib_) dlclose(lib_);
}
}  // namespace volpath
