// halo.hpp -- ghost-cell halo exchange between per-GPU blocks over RCCL (xGMI).
//
// Replaces PETSc DMDA globalToLocal (reference src/petclaw/state.py:254-269: BOX stencil,
// width mbc, faces AND corners) and Vec.max (src/petclaw/cfl.py:29-31).
//
// One block per process/GPU.  Per exchange: pack the <= 8 outgoing strips into contiguous
// device buffers (one kernel), one ncclGroup of Send/Recv pairs (point-to-point: every 2x4
// neighbour on an 8-GPU xGMI node is one hop), unpack the incoming strips into the ghost
// frame (one kernel).  Messages are tiny (<= 328 KB at 8192^2 on 2x4) => latency bound;
// everything is enqueued on the caller's stream, no host synchronisation.
//
// librccl is dlopen()ed on first use so that single-GPU runs never load it.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string>

namespace pcl {

// direction order is part of the wire protocol (send order == matching recv order)
enum Dir { W = 0, E = 1, S = 2, N = 3, SW = 4, SE = 5, NW = 6, NE = 7 };
__host__ __device__ inline int opposite(int d) {
    const int o[8] = {E, W, N, S, NE, NW, SE, SW};
    return o[d];
}

struct HaloRegion { int i0, j0, ni, nj; };

// strip to SEND towards direction d (interior cells next to that edge) when send=true,
// ghost strip to FILL from direction d when send=false
// gy: ghost width of the second index (default = g; 0 for a 1-D grid, whose single row has no ghost rows: only the
// W and E strips exist then)
__host__ __device__ inline HaloRegion halo_region(int d, bool send, int I, int J, int g, int gy = -1) {
    if (gy < 0) gy = g;
    const int mx = I - 2 * g, my = J - 2 * gy;
    int xs, xn, ys, yn;  // x start/len, y start/len
    const bool west = (d == W || d == SW || d == NW), east = (d == E || d == SE || d == NE);
    const bool south = (d == S || d == SW || d == SE), north = (d == N || d == NW || d == NE);
    if (west) { xs = send ? g : 0; xn = g; }
    else if (east) { xs = send ? I - 2 * g : I - g; xn = g; }
    else { xs = g; xn = mx; }
    if (south) { ys = send ? gy : 0; yn = gy; }
    else if (north) { ys = send ? J - 2 * gy : J - gy; yn = gy; }
    else { ys = gy; yn = my; }
    return HaloRegion{xs, ys, xn, yn};
}

struct HaloPlan {
    int nbr[8];
    long off[8];  // offset (doubles, per component count 1) of each direction's buffer
    long cnt[8];  // doubles per component in each direction's strip (cells x elem)
    int I, J, g;  // the decomposed index plane (with ghosts) and the ghost width
    int gy;       // ghost width of the plane's second index (= g; 0 for 1-D grids)
    // 2-D blocks: the plane is (i, j), one cell per element: elem = 1, pi = 1, pj = pitch.
    // 3-D blocks decomposed over (y, z): the plane is (j, k) and every element is a whole x-row of the
    // array (elem = cells per row incl. ghosts, contiguous): pi = pitch, pj = pitch * J3.
    int elem;
    long pi, pj;
};

__global__ void halo_pack(const double *q, double *buf, HaloPlan p, int nm, long pitch, long plane,
                          bool unpack) {
    const int d = blockIdx.y;
    if (p.nbr[d] < 0) return;
    const HaloRegion r = halo_region(d, !unpack, p.I, p.J, p.g, p.gy);
    const long ncell = (long)r.ni * r.nj * p.elem;
    (void)pitch;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < ncell * nm;
         t += (long)gridDim.x * blockDim.x) {
        const int m = (int)(t / ncell);
        const long c = t % ncell;
        const int e = (int)(c % p.elem);
        const long cc = c / p.elem;
        const int jj = (int)(cc / r.ni), ii = (int)(cc % r.ni);
        const long g = m * plane + (long)(r.j0 + jj) * p.pj + (long)(r.i0 + ii) * p.pi + e;
        const long b = p.off[d] * nm + t;
        if (unpack) const_cast<double *>(q)[g] = buf[b];
        else buf[b] = q[g];
    }
}

class Halo {
public:
    bool active = false;

    static int load(std::string &err) {
        if (api().ok) return 0;
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { err = std::string("dlopen librccl: ") + dlerror(); return -1; }
        Api &a = api();
#define PCL_SYM(name)                                                      \
    a.name = (decltype(a.name))dlsym(h, #name);                            \
    if (!a.name) { err = "librccl lacks " #name; return -1; }
        PCL_SYM(ncclGetUniqueId) PCL_SYM(ncclCommInitRank) PCL_SYM(ncclCommDestroy)
        PCL_SYM(ncclSend) PCL_SYM(ncclRecv) PCL_SYM(ncclGroupStart) PCL_SYM(ncclGroupEnd)
        PCL_SYM(ncclAllReduce) PCL_SYM(ncclGetErrorString)
#undef PCL_SYM
        a.ok = true;
        return 0;
    }

    static int unique_id(char uid[128], std::string &err) {
        if (load(err)) return -1;
        ncclUniqueId id;
        ncclResult_t r = api().ncclGetUniqueId(&id);
        if (r != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + api().ncclGetErrorString(r); return -1; }
        memcpy(uid, id.internal, 128);
        return 0;
    }

    int init(int nranks, int rank, const char uid[128], const int nbr[8], int I, int J, int g,
             int nmax, hipStream_t stream, std::string &err, int elem = 1, long pi = 1, long pj = 0, int gy = -1) {
        if (load(err)) return -1;
        destroy();
        stream_ = stream;
        plan_.I = I; plan_.J = J; plan_.g = g; plan_.gy = gy < 0 ? g : gy;
        plan_.elem = elem; plan_.pi = pi; plan_.pj = pj;
        long off = 0;
        for (int d = 0; d < 8; d++) {
            if (nbr[d] >= nranks) { err = "neighbour rank out of range"; return -1; }
            plan_.nbr[d] = nbr[d];
            const HaloRegion r = halo_region(d, true, I, J, g, plan_.gy);
            plan_.cnt[d] = (long)r.ni * r.nj * elem;
            plan_.off[d] = off;
            off += plan_.cnt[d];
        }
        cells_ = off;
        const size_t bytes = (size_t)cells_ * nmax * sizeof(double);
        if (hipMalloc((void **)&send_, bytes) != hipSuccess || hipMalloc((void **)&recv_, bytes) != hipSuccess ||
            hipMalloc((void **)&red_, 64) != hipSuccess ||
            hipHostMalloc((void **)&red_host_, 64, hipHostMallocDefault) != hipSuccess) {
            err = "halo buffer allocation failed";
            return -1;
        }
        ncclUniqueId id;
        memcpy(id.internal, uid, 128);
        ncclResult_t r = api().ncclCommInitRank(&comm_, nranks, id, rank);
        if (r != ncclSuccess) { err = std::string("ncclCommInitRank: ") + api().ncclGetErrorString(r); return -1; }
        active = true;
        return 0;
    }

    // ---- host-staged transport (diagnostics / multi-process tests on ONE device) ---------------------------
    // Same plan, same pack / unpack kernels, same call sites and stream choreography; only the wire differs: the
    // packed strips go to pinned host memory, a callback of the embedding program carries them between the
    // processes (pyclaw_amd.parallel: its TCP group), and the received strips go back up.  RCCL refuses two ranks on
    // one device, so this is how 2-6 ranks sharing one GPU exercise the whole decomposed device path.  Blocking.
    typedef int (*ExchangeFn)(void *user, const double *send, double *recv, const long *off, const long *cnt,
                              const int *nbr, int nm);
    typedef int (*ReduceFn)(void *user, double *value);
    int init_host(int nranks, int rank, const int nbr[8], int I, int J, int g, int nmax, hipStream_t stream,
                  ExchangeFn xfn, ReduceFn rfn, void *user, std::string &err, int elem = 1, long pi = 1, long pj = 0, int gy = -1) {
        destroy();
        (void)rank;
        stream_ = stream;
        plan_.I = I; plan_.J = J; plan_.g = g; plan_.gy = gy < 0 ? g : gy;
        plan_.elem = elem; plan_.pi = pi; plan_.pj = pj;
        long off = 0;
        for (int d = 0; d < 8; d++) {
            if (nbr[d] >= nranks) { err = "neighbour rank out of range"; return -1; }
            plan_.nbr[d] = nbr[d];
            const HaloRegion r = halo_region(d, true, I, J, g, plan_.gy);
            plan_.cnt[d] = (long)r.ni * r.nj * elem;
            plan_.off[d] = off;
            off += plan_.cnt[d];
        }
        cells_ = off;
        const size_t bytes = (size_t)cells_ * nmax * sizeof(double);
        if (hipMalloc((void **)&send_, bytes) != hipSuccess || hipMalloc((void **)&recv_, bytes) != hipSuccess ||
            hipMalloc((void **)&red_, 64) != hipSuccess ||
            hipHostMalloc((void **)&red_host_, 64, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&hsend_, bytes, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&hrecv_, bytes, hipHostMallocDefault) != hipSuccess) {
            err = "halo buffer allocation failed";
            return -1;
        }
        xfn_ = xfn; rfn_ = rfn; user_ = user;
        active = true;
        return 0;
    }

    bool has(int d) const { return active && plan_.nbr[d] >= 0; }

    // exchange the ghost frame of an nm-component SoA array
    // `on`: stream to enqueue on (default: the solver stream given to init)
    int exchange(double *q, int nm, long pitch, long plane, std::string &err, hipStream_t on = nullptr) {
        if (!active) { err = "halo exchange before pcl_comm_init"; return -1; }
        Api &a = api();
        hipStream_t stream_ = on ? on : this->stream_;
        dim3 grid(64, 8);
        hipLaunchKernelGGL(halo_pack, grid, dim3(256), 0, stream_, q, send_, plan_, nm, pitch, plane, false);
        if (xfn_) {
            const size_t bytes = (size_t)cells_ * nm * sizeof(double);
            if (hipMemcpyAsync(hsend_, send_, bytes, hipMemcpyDeviceToHost, stream_) != hipSuccess ||
                hipStreamSynchronize(stream_) != hipSuccess) { err = "halo D2H failed"; return -1; }
            if (xfn_(user_, hsend_, hrecv_, plan_.off, plan_.cnt, plan_.nbr, nm)) { err = "host halo transport failed"; return -1; }
            if (hipMemcpyAsync(recv_, hrecv_, bytes, hipMemcpyHostToDevice, stream_) != hipSuccess) { err = "halo H2D failed"; return -1; }
            hipLaunchKernelGGL(halo_pack, grid, dim3(256), 0, stream_, q, recv_, plan_, nm, pitch, plane, true);
            if (hipGetLastError() != hipSuccess) { err = "halo pack/unpack launch failed"; return -1; }
            return 0;
        }
        ncclResult_t r = a.ncclGroupStart();
        for (int d = 0; d < 8 && r == ncclSuccess; d++) {
            // what I send towards d fills the receiver's ghost strip on its opposite(d) side
            if (plan_.nbr[d] >= 0)
                r = a.ncclSend(send_ + plan_.off[d] * nm, (size_t)plan_.cnt[d] * nm, ncclDouble,
                               plan_.nbr[d], comm_, stream_);
            const int o = opposite(d);
            if (r == ncclSuccess && plan_.nbr[o] >= 0)
                r = a.ncclRecv(recv_ + plan_.off[o] * nm, (size_t)plan_.cnt[o] * nm, ncclDouble,
                               plan_.nbr[o], comm_, stream_);
        }
        ncclResult_t r2 = a.ncclGroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) { err = std::string("halo send/recv: ") + a.ncclGetErrorString(r); return -1; }
        hipLaunchKernelGGL(halo_pack, grid, dim3(256), 0, stream_, q, recv_, plan_, nm, pitch, plane, true);
        if (hipGetLastError() != hipSuccess) { err = "halo pack/unpack launch failed"; return -1; }
        return 0;
    }

    // in-place max all-reduce of one double that already lives on the device (the CFL word),
    // enqueued on the solver stream: no extra host round trip
    int allreduce_max_device(double *dev, std::string &err) {
        if (!active) return 0;
        if (rfn_) {
            if (hipMemcpyAsync(red_host_, dev, sizeof(double), hipMemcpyDeviceToHost, stream_) != hipSuccess ||
                hipStreamSynchronize(stream_) != hipSuccess) { err = "allreduce D2H failed"; return -1; }
            if (rfn_(user_, red_host_)) { err = "host allreduce failed"; return -1; }
            if (hipMemcpyAsync(dev, red_host_, sizeof(double), hipMemcpyHostToDevice, stream_) != hipSuccess) {
                err = "allreduce H2D failed"; return -1;
            }
            return 0;
        }
        ncclResult_t r = api().ncclAllReduce(dev, dev, 1, ncclDouble, ncclMax, comm_, stream_);
        if (r != ncclSuccess) { err = std::string("ncclAllReduce: ") + api().ncclGetErrorString(r); return -1; }
        return 0;
    }

    int allreduce_max(double *v, std::string &err) {
        if (!active) { err = "allreduce before pcl_comm_init"; return -1; }
        if (rfn_) {
            if (rfn_(user_, v)) { err = "host allreduce failed"; return -1; }
            return 0;
        }
        Api &a = api();
        *red_host_ = *v;
        if (hipMemcpyAsync(red_, red_host_, sizeof(double), hipMemcpyHostToDevice, stream_) != hipSuccess) {
            err = "allreduce H2D failed"; return -1;
        }
        ncclResult_t r = a.ncclAllReduce(red_, red_, 1, ncclDouble, ncclMax, comm_, stream_);
        if (r != ncclSuccess) { err = std::string("ncclAllReduce: ") + a.ncclGetErrorString(r); return -1; }
        if (hipMemcpyAsync(red_host_, red_, sizeof(double), hipMemcpyDeviceToHost, stream_) != hipSuccess ||
            hipStreamSynchronize(stream_) != hipSuccess) {
            err = "allreduce D2H failed"; return -1;
        }
        *v = *red_host_;
        return 0;
    }

    void destroy() {
        if (comm_) { api().ncclCommDestroy(comm_); comm_ = nullptr; }
        if (send_) hipFree(send_);
        if (recv_) hipFree(recv_);
        if (red_) hipFree(red_);
        if (red_host_) hipHostFree(red_host_);
        if (hsend_) hipHostFree(hsend_);
        if (hrecv_) hipHostFree(hrecv_);
        send_ = recv_ = red_ = nullptr; red_host_ = hsend_ = hrecv_ = nullptr;
        xfn_ = nullptr; rfn_ = nullptr; user_ = nullptr;
        active = false;
    }

private:
    struct Api {
        bool ok = false;
        decltype(&::ncclGetUniqueId) ncclGetUniqueId = nullptr;
        decltype(&::ncclCommInitRank) ncclCommInitRank = nullptr;
        decltype(&::ncclCommDestroy) ncclCommDestroy = nullptr;
        decltype(&::ncclSend) ncclSend = nullptr;
        decltype(&::ncclRecv) ncclRecv = nullptr;
        decltype(&::ncclGroupStart) ncclGroupStart = nullptr;
        decltype(&::ncclGroupEnd) ncclGroupEnd = nullptr;
        decltype(&::ncclAllReduce) ncclAllReduce = nullptr;
        decltype(&::ncclGetErrorString) ncclGetErrorString = nullptr;
    };
    static Api &api() { static Api a; return a; }

    ncclComm_t comm_ = nullptr;
    hipStream_t stream_ = nullptr;
    HaloPlan plan_{};
    long cells_ = 0;
    double *send_ = nullptr, *recv_ = nullptr, *red_ = nullptr, *red_host_ = nullptr;
    double *hsend_ = nullptr, *hrecv_ = nullptr;     // host-staged transport
    ExchangeFn xfn_ = nullptr;
    ReduceFn rfn_ = nullptr;
    void *user_ = nullptr;
};

}  // namespace pcl
