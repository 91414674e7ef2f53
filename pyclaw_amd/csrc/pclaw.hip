// pclaw.hip -- C ABI of libpyclaw_amd.so (see include/pyclaw_amd.h).
// gfx950 only.  No CPU fallback: every entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <chrono>
#include <string>
#include <utility>
#include <vector>
#include <csignal>
#include <execinfo.h>

#include "../../include/pyclaw_amd.h"
#include "sweep_args.hpp"
#include "halo.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(PCL_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

}  // namespace

struct pcl_solver {
    pcl_config cfg;
    int I = 1, J = 1, K = 1;  // cells incl. ghosts (K > 1 only in 3-D: q[m][k][j][i], row r = k*J + j)
    long pitch = 0, plane = 0, total = 0;  // doubles
    long aplane = 0;
    int lead = 0;         // doubles the array bases are shifted by so that the first interior
                          // cell of every row starts a 128-byte line (16 - mbc)
    double *q = nullptr, *t1 = nullptr, *t2 = nullptr, *t3 = nullptr, *bak = nullptr;
    double *aux = nullptr;
    double *sreg[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // SharpClaw registers (0 aliases q)
    int sel = 0;          // register the put/get/bc/strip/halo calls act on
    double *scr3[14] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                        nullptr, nullptr, nullptr};   // unsplit 3-D slice pieces
    double *stage = nullptr;  // AoS staging for host transfers (qbc-sized)
    size_t stage_bytes = 0;
    unsigned long long *cfl_dev = nullptr;
    unsigned long long *cfl_host = nullptr;  // pinned, device-visible: [0] = value, [1] = sequence number
    unsigned long long *cfl_host_dev = nullptr;  // the same words as the device sees them
    unsigned long long cfl_seq = 0;
    int cfl_poll = 1;     // PCL_CFL_POLL=0: D2H copy + event wait instead of the mapped word
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_cfl = nullptr;
    double **undo_slot = nullptr;  // buffer that holds the pre-step state
    int fused_src = 0;    // source term applied inside the last pass of the dim-split step (pcl_fuse_source)
    double fused_src_p[2] = {0.0, 0.0};
    int timing = 0;       // 0 off, N >= 1: HIP events around the sweep launches of every N-th step (pcl_kernel_timing)
    long step_no = 0;     // hyperbolic steps / stages completed (read_cfl)
    int vbc_on = 0;       // next x pass evaluates these BCs while loading (pcl_bc_step)
    int vbc[4] = {-1, -1, -1, -1};
    double vconst[4][8] = {};
    struct Timed { hipEvent_t a, b; int which; bool count; };
    std::vector<Timed> timed;
    std::vector<hipEvent_t> evpool;
    double kt_ms[3] = {0, 0, 0};      // x pass / y pass (or phases) / the one-kernel dim-split step
    long kt_n[3] = {0, 0, 0};
    // Which form of the dimension-split 2-D step runs (identical results): PCL_TUNE_FUSED_STEP = 0 two passes, 1 one
    // kernel, 2 (default) the faster of the two, measured: 64 steps into every window of FORM_WINDOW steps a few steps
    // of each form are timed on the host (a step ends with the Courant number's read-back) and the rest of the window
    // runs the faster one.  The one-kernel step halves the HBM traffic and wins wherever most wavefronts take the
    // no-jump shortcut; where every cell is active its halo rows (32 / 28 of the arithmetic) make the two passes faster.
    int form_seq_tune = 0;          // set by pcl_bc_step around the sequential decomposed step: the trials run there too
    int form_now = 1;               // 1 = one kernel, 0 = two passes
    long form_step = 0;             // steps since the window began
    double form_t[2] = {0, 0};      // best (smallest) wall time of a trial step of each form in this window
    long form_steps[2] = {0, 0};    // steps run in each form (pcl_step_form_stats)
    pcl::Halo halo;
    // halo exchange overlapped with the interior of the x pass (pcl_bc_step, dim-split 2-D)
    hipStream_t hstream = nullptr;
    hipEvent_t ev_h0 = nullptr, ev_h1 = nullptr;
    int overlap = 1;
    bool overlap_dflt = true;       // PCL_HALO_OVERLAP not set: the library picks per kernel family (twopass_overlap_ok)
    // Exchange-ahead (pcl_halo_exchange_ahead): the halo exchange of the NEW state is enqueued on the halo stream
    // right behind the y pass that produced it, so it runs through the hand-over of the Courant number, the host's
    // accept / retake decision and the first tiles of the next x pass instead of in front of that pass' rim tiles.
    // ghost_ok[] names the buffers whose ghost frame such an exchange (or the one at the start of a step) has filled
    // and nothing has touched since: the state itself and, for a retaken step, the pre-step state (pcl_undo_step).
    int exchange_ahead = 0;
    hipEvent_t ev_y = nullptr;
    const double *ghost_ok[2] = {nullptr, nullptr};
    bool ghosts_filled(const double *buf) const { return buf && (ghost_ok[0] == buf || ghost_ok[1] == buf); }
    void ghosts_mark(const double *buf) { if (!ghosts_filled(buf)) { ghost_ok[1] = ghost_ok[0]; ghost_ok[0] = buf; } }
    void ghosts_drop(const double *buf) { for (auto &g : ghost_ok) if (g == buf) g = nullptr; }
    void ghosts_drop_all() { ghost_ok[0] = ghost_ok[1] = nullptr; }
};

static inline double *&cur(pcl_solver *s) { return s->sel == 0 ? s->q : s->sreg[s->sel]; }
static inline bool timing_on(const pcl_solver *s) { return s->timing > 0 && s->step_no % s->timing == 0; }

namespace pcl { namespace exact { void launch_shift_test(const double *in, double *l, double *r); } }
// one launcher per arithmetic mode (kernels.hip is compiled once per mode)
#define PCL_BY_MATH(math, call) \
    ((math) == PCL_MATH_FAST ? pcl::fast::call : (math) == PCL_MATH_STRICT ? pcl::strict::call : pcl::exact::call)

namespace {

using namespace pcl;

// ---- layout conversion kernels: host Fortran AoS (m fastest) <-> device SoA planes ----
// hostlike[(m) + meqn*(i + ni*j)] where (i,j) run over a window [io,io+ni) x [jo,jo+nj)
// 3-D: blockIdx.z = k of the window, ko its offset, slab = doubles between k-planes (J*pitch).
__global__ void aos_to_soa(const double *__restrict__ src, double *__restrict__ dst, int nm, int ni,
                           int nj, int io, int jo, long pitch, long plane, int ko = 0, long slab = 0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, k = blockIdx.z;
    if (i >= ni || j >= nj) return;
    const long h = (long)nm * (i + (long)ni * (j + (long)nj * k));
    const long d = (long)(k + ko) * slab + (long)(j + jo) * pitch + (i + io);
    for (int m = 0; m < nm; m++) dst[m * plane + d] = src[h + m];
}
__global__ void soa_to_aos(const double *__restrict__ src, double *__restrict__ dst, int nm, int ni,
                           int nj, int io, int jo, long pitch, long plane, int ko = 0, long slab = 0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y, k = blockIdx.z;
    if (i >= ni || j >= nj) return;
    const long h = (long)nm * (i + (long)ni * (j + (long)nj * k));
    const long d = (long)(k + ko) * slab + (long)(j + jo) * pitch + (i + io);
    for (int m = 0; m < nm; m++) dst[h + m] = src[m * plane + d];
}

// ---- ghost-cell fills, solver.py:384-452 ---------------------------------------------------
// One thread per (transverse index t, ghost layer g, component m).
__global__ void bc_kernel(double *q, int nm, int I, int J, long pitch, long plane, int mbc, int idim,
                          int side, int type, pcl::RpParams cstate, int is_aux) {
    const int nt = idim == 0 ? J : I;
    const int N = idim == 0 ? I : J;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)nt * mbc;
    if (tid >= per * nm) return;
    const int m = (int)(tid / per);
    const int rem = (int)(tid % per);
    const int g = rem / nt;  // ghost layer 0..mbc-1
    const int t = rem % nt;
    int dstk, srck;
    double sign = 1.0;
    if (side == 0) {
        dstk = g;
        if (type == PCL_BC_OUTFLOW) srck = mbc;
        else if (type == PCL_BC_PERIODIC) srck = N - 2 * mbc + g;
        else { srck = 2 * mbc - 1 - g; if (type == PCL_BC_REFLECTING && m == idim + 1 && !is_aux) sign = -1.0; }
    } else {
        dstk = N - 1 - g;
        if (type == PCL_BC_OUTFLOW) srck = N - mbc - 1;
        else if (type == PCL_BC_PERIODIC) srck = 2 * mbc - 1 - g;  // q[N-mbc+k] = q[mbc+k], k=mbc-1-g
        else { srck = N - 2 * mbc + g; if (type == PCL_BC_REFLECTING && m == idim + 1 && !is_aux) sign = -1.0; }
    }
    const long d = idim == 0 ? (long)t * pitch + dstk : (long)dstk * pitch + t;
    if (type == 100) {  // constant inflow state
        q[m * plane + d] = cstate.v[m];
        return;
    }
    const long s = idim == 0 ? (long)t * pitch + srck : (long)srck * pitch + t;
    const double v = q[m * plane + s];
    q[m * plane + d] = (sign < 0.0) ? -v : v;
}

// 3-D ghost fill: one thread per (component m, ghost layer g, transverse cell t) of side `side` of dimension
// idim; n[] = extents with ghosts, st[] = strides in doubles.  Same rules as bc_kernel (solver.py:384-452).
struct Dims3 { int n[3]; long st[3]; };
__global__ void bc3_kernel(double *q, int nm, Dims3 D, long plane, int mbc, int idim, int side, int type,
                           int is_aux) {
    const int d1 = idim == 0 ? 1 : 0, d2 = idim == 2 ? 1 : 2;  // the two transverse dimensions
    const long nt = (long)D.n[d1] * D.n[d2];
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nt * mbc * nm) return;
    const int m = (int)(tid / (nt * mbc));
    const long rem = tid % (nt * mbc);
    const int g = (int)(rem / nt);
    const long t = rem % nt;
    const int t1 = (int)(t % D.n[d1]), t2 = (int)(t / D.n[d1]);
    const int N = D.n[idim];
    int dstk, srck;
    bool neg = false;
    if (side == 0) {
        dstk = g;
        if (type == PCL_BC_OUTFLOW) srck = mbc;
        else if (type == PCL_BC_PERIODIC) srck = N - 2 * mbc + g;
        else { srck = 2 * mbc - 1 - g; neg = (m == idim + 1 && !is_aux); }
    } else {
        dstk = N - 1 - g;
        if (type == PCL_BC_OUTFLOW) srck = N - mbc - 1;
        else if (type == PCL_BC_PERIODIC) srck = 2 * mbc - 1 - g;
        else { srck = N - 2 * mbc + g; neg = (m == idim + 1 && !is_aux); }
    }
    const long tr = (long)t1 * D.st[d1] + (long)t2 * D.st[d2];
    const double v = q[m * plane + tr + (long)srck * D.st[idim]];
    q[m * plane + tr + (long)dstk * D.st[idim]] = neg ? -v : v;
}

// ---- Euler radial source, test/euler/2d/shockbubble.py:59-94 -------------------------------
__global__ void src_euler_radial(double *q, const double *aux, int mbc, int mx, int my, long pitch,
                                 long plane, double dt, double gamma1, double ndm1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= mx || j >= my) return;
    const long g = (long)(j + mbc) * pitch + (i + mbc);
    double q0 = q[g], q1 = q[plane + g], q2 = q[2 * plane + g], q3 = q[3 * plane + g];
    pcl::euler_radial_source(q0, q1, q2, q3, aux[g], dt, gamma1, ndm1);
    q[g] = q0;
    q[plane + g] = q1;
    q[2 * plane + g] = q2;
    q[3 * plane + g] = q3;
}

// ---- sphere app: custom y boundary, shallow_4_Rossby_Haurwitz_wave.py:295-313 ---------------------------
// lower: qbc[:, i, j] = qbc[:, I-1-i, 2*mbc-1-j]; upper: qbc[:, i, J-mbc+j] = qbc[:, I-1-i, J-mbc-1-j]   (j < mbc, all i)
__global__ void bc_sphere_mirror(double *q, int nm, int I, int J, long pitch, long plane, int mbc, int side) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)I * mbc;
    if (tid >= per * nm) return;
    const int m = (int)(tid / per), rem = (int)(tid % per), j = rem / I, i = rem % I;
    const int dst = side == 0 ? j : J - mbc + j;
    const int src = side == 0 ? 2 * mbc - 1 - j : J - mbc - 1 - j;
    q[m * plane + (long)dst * pitch + i] = q[m * plane + (long)src * pitch + (I - 1 - i)];
}

// ---- sphere app: Coriolis source, apps/shallow-sphere/src2.f:43-146 -----------------------------------------
// Per interior cell: project the momentum onto the tangent plane, 4-stage Runge-Kutta on the Coriolis term,
// project again.  src2.f takes the radial vector of the Coriolis part from mapc2p(cell centre): the very values
// setaux.f:150-156 stored in aux(14:16) (same expression, same arguments).  The Fortran does the three phases in
// three loops over the grid; cells are independent, so one pass per cell gives the same numbers.
__global__ void src_sphere_coriolis(double *q, const double *aux, int mbc, int mx, int my, long pitch, long plane,
                                    double dt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= mx || j >= my) return;
    const long g = (long)(j + mbc) * pitch + (i + mbc);
    const double df = (double)12.600576f;            // src2.f:39: a REAL*4 literal
    const double erx = aux[13 * plane + g], ery = aux[14 * plane + g], erz = aux[15 * plane + g];
    double q2 = q[plane + g], q3 = q[2 * plane + g], q4 = q[3 * plane + g];
    double qn = erx * q2 + ery * q3 + erz * q4;
    q2 = q2 - qn * erx; q3 = q3 - qn * ery; q4 = q4 - qn * erz;
    const double fcor = df * erz;
    double hu = q2, hv = q3, hw = q4, k1[3], k2[3], k3[3], k4[3];
    k1[0] = fcor * dt * (erz * hv - ery * hw); k1[1] = dt * fcor * (erx * hw - erz * hu); k1[2] = dt * fcor * (ery * hu - erx * hv);
    hu = q2 + 0.5 * k1[0]; hv = q3 + 0.5 * k1[1]; hw = q4 + 0.5 * k1[2];
    k2[0] = fcor * dt * (erz * hv - ery * hw); k2[1] = dt * fcor * (erx * hw - erz * hu); k2[2] = dt * fcor * (ery * hu - erx * hv);
    hu = q2 + 0.5 * k2[0]; hv = q3 + 0.5 * k2[1]; hw = q4 + 0.5 * k2[2];
    k3[0] = fcor * dt * (erz * hv - ery * hw); k3[1] = dt * fcor * (erx * hw - erz * hu); k3[2] = dt * fcor * (ery * hu - erx * hv);
    hu = q2 + 0.5 * k3[0]; hv = q3 + 0.5 * k3[1]; hw = q4 + 0.5 * k3[2];
    k4[0] = fcor * dt * (erz * hv - ery * hw); k4[1] = dt * fcor * (erx * hw - erz * hu); k4[2] = dt * fcor * (ery * hu - erx * hv);
    q2 = q2 + (k1[0] + 2.0 * k2[0] + 2.0 * k3[0] + k4[0]) / 6.0;
    q3 = q3 + (k1[1] + 2.0 * k2[1] + 2.0 * k3[1] + k4[1]) / 6.0;
    q4 = q4 + (k1[2] + 2.0 * k2[2] + 2.0 * k3[2] + k4[2]) / 6.0;
    qn = erx * q2 + ery * q3 + erz * q4;
    q[plane + g] = q2 - qn * erx;
    q[2 * plane + g] = q3 - qn * ery;
    q[3 * plane + g] = q4 - qn * erz;
}

// ---- sweep dispatch ---------------------------------------------------------------------------
SweepArgs make_args(pcl_solver *s, const double *qin, double *qout, int ids, double dt) {
    SweepArgs a;
    memset(&a, 0, sizeof(a));
    a.qin = qin;
    a.qout = qout;
    a.aux = s->aux;
    a.pitch = s->pitch;
    a.plane = s->plane;
    a.I = s->I;
    a.J = s->J;
    a.mbc = s->cfg.mbc;
    a.mx = s->cfg.n[0];
    a.my = s->cfg.ndim > 1 ? s->cfg.n[1] : 1;
    a.mcapa = s->cfg.method[5];
    a.order = s->cfg.method[1];
    for (int k = 0; k < PCL_MAX_WAVES; k++) a.mthlim[k] = s->cfg.mthlim[k];
    a.dt = dt;
    a.dx = s->cfg.d[ids - 1];
    a.dtd = dt / s->cfg.d[ids - 1];
    for (int k = 0; k < PCL_MAX_RP_PARAMS; k++) a.par.v[k] = s->cfg.rp_params[k];
    a.cfl = s->cfl_dev;
    a.vbc_on = (ids == 1) ? s->vbc_on : 0;
    // the fused source belongs to the LAST pass of the dimension-split 2-D step (pcl_fuse_source)
    a.src_id = (ids == 2 && s->cfg.ndim == 2) ? s->fused_src : 0;      // y pass (dim-split) / y phase (unsplit)
    a.src_p[0] = s->fused_src_p[0]; a.src_p[1] = s->fused_src_p[1];
    for (int k = 0; k < 4; k++) { a.vbc[k] = s->vbc[k]; for (int m = 0; m < 8; m++) a.vconst[k][m] = s->vconst[k][m]; }
    static const int ablate = [] { const char *e = getenv("PCL_TUNE_ABLATE"); return e ? atoi(e) : 0; }();
    a.ablate = ablate;
    static const int xcd = [] { const char *e = getenv("PCL_TUNE_XCD"); return e ? atoi(e) : 5; }();
    a.xcd = xcd;
    return a;
}

hipEvent_t get_event(pcl_solver *s) {
    if (!s->evpool.empty()) {
        hipEvent_t e = s->evpool.back();
        s->evpool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

int drain_timing(pcl_solver *s) {
    if (s->timed.empty()) return PCL_OK;
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (auto &t : s->timed) {
        float ms = 0;
        hipEventElapsedTime(&ms, t.a, t.b);
        s->kt_ms[t.which] += ms;
        s->kt_n[t.which] += t.count ? 1 : 0;
        s->evpool.push_back(t.a);
        s->evpool.push_back(t.b);
    }
    s->timed.clear();
    return PCL_OK;
}

// Ghost frame of the unsplit step in ONE launch: every cell outside the interior gets its boundary value -- the
// composition of the per-side fills of solver.py:354-452 (x sides first, then y sides over the x-filled columns, so a
// corner cell is the y rule applied to an x-ghost cell) written as an index remap, like the dim-split x pass does
// while loading its tiles -- and is copied to dst (the y phase updates t1 in place and its ghost frame must equal
// qold's).  bc[k] < 0: no fill on that side (neighbour block or a fill done elsewhere); 100 = constant state.
// only: 0 = every ghost cell; 1 / 2 = only the ghost columns / rows beyond the two layers the sweep tiles hold (mbc > 2:
// the cells no tile of an x / y pass writes, so the copy may run beside the tiles)
struct FrameBc { int t[4]; double c[4][8]; int only; };
static inline unsigned frame_blocks(const pcl_solver *s) {      // one thread per ghost cell, at most 256 workgroups
    const long n = s->J > 2 * s->cfg.mbc ? 2L * s->cfg.mbc * (s->I + s->J - 2 * s->cfg.mbc) : (long)s->I * s->J;
    const long b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 256 ? 256 : b));
}
__device__ __forceinline__ void frame_map(int k, int n, int mbc, int lo, int hi, int &src, bool &neg, bool &cst, int &side) {
    src = k; neg = false; cst = false; side = 0;
    if (k < mbc && lo >= 0) {
        if (lo == PCL_BC_OUTFLOW) src = mbc;
        else if (lo == PCL_BC_PERIODIC) src = n - 2 * mbc + k;
        else if (lo == PCL_BC_REFLECTING) { src = 2 * mbc - 1 - k; neg = true; }
        else cst = true;
    } else if (k >= n - mbc && hi >= 0) {
        side = 1;
        if (hi == PCL_BC_OUTFLOW) src = n - mbc - 1;
        else if (hi == PCL_BC_PERIODIC) src = k - (n - 2 * mbc);
        else if (hi == PCL_BC_REFLECTING) { src = 2 * (n - mbc) - 1 - k; neg = true; }
        else cst = true;
    }
}
__global__ void frame_kernel(double *q, double *dst, int nm, int I, int J, int mbc, long pitch, long plane, FrameBc f) {
    // only the ghost cells are enumerated: 2*mbc whole rows (bottom, top), then 2*mbc cells of every other row.
    // (1-D grids, J == 1, have no ghost rows: every cell of the single row is visited; interior ones map to themselves.)
    const bool rows = J > 2 * mbc;
    const long nfull = rows ? 2L * mbc * I : 0;
    const long ncell = rows ? nfull + 2L * mbc * (J - 2 * mbc) : (long)I * J;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < ncell; t += (long)gridDim.x * blockDim.x) {
        int i, j;
        if (!rows) { i = (int)(t % I); j = (int)(t / I); }
        else if (t < nfull) { const int r = (int)(t / I); i = (int)(t % I); j = r < mbc ? r : J - 2 * mbc + r; }
        else { const long u = t - nfull; const int c = (int)(u % (2 * mbc)); j = mbc + (int)(u / (2 * mbc)); i = c < mbc ? c : I - 2 * mbc + c; }
        if (i >= mbc && i < I - mbc && j >= mbc && j < J - mbc) continue;
        if (f.only == 1 && i >= mbc - 2 && i < I - (mbc - 2)) continue;
        if (f.only == 2 && j >= mbc - 2 && j < J - (mbc - 2)) continue;
        int si, sj, sdi, sdj;
        bool ni, nj, ci, cj;
        // y first (the side applied LAST): the sphere app's pole boundary also reverses the row over its whole ghosted
        // width (bc_sphere_mirror), so the x rule then applies to the reversed column
        int ix = i;
        const bool mir_lo = j < mbc && f.t[2] == PCL_BC_SPHERE_MIRROR, mir_hi = j >= J - mbc && f.t[3] == PCL_BC_SPHERE_MIRROR;
        if (mir_lo || mir_hi) {
            sj = mir_lo ? 2 * mbc - 1 - j : 2 * (J - mbc) - 1 - j;
            nj = false; cj = false; sdj = mir_lo ? 0 : 1;
            ix = I - 1 - i;
        } else
            frame_map(j, J, mbc, f.t[2], f.t[3], sj, nj, cj, sdj);
        frame_map(ix, I, mbc, f.t[0], f.t[1], si, ni, ci, sdi);
        const long g = (long)j * pitch + i, gs = (long)sj * pitch + si;
        for (int m = 0; m < nm; m++) {
            double v = q[m * plane + gs];
            if (m == 1) v = ni ? -v : v;
            v = ci ? f.c[sdi][m < 8 ? m : 7] : v;
            if (m == 2) v = nj ? -v : v;
            v = cj ? f.c[2 + sdj][m < 8 ? m : 7] : v;
            if (gs != g || ci || cj) q[m * plane + g] = v;      // (a constant-state cell maps to itself)
            if (dst) dst[m * plane + g] = v;
        }
    }
}

// one directional sweep qin -> qout; ids 1 = x (or the 1-D step), 2 = y
// sub/box: tile subset of the x pass (sweep_args.hpp); the second launch of a split pass adds its time to
// the pass without counting as another launch
int do_sweep(pcl_solver *s, const double *qin, double *qout, int ids, double dt, int sub = 0,
             const int *box = nullptr, hipStream_t on = nullptr) {
    hipStream_t stream = on ? on : s->stream;
    SweepArgs a = make_args(s, qin, qout, ids, dt);
    a.sub = sub;
    if (sub) for (int k = 0; k < 4; k++) a.box[k] = box[k];
    if (s->cfg.mbc > 2 && s->cfg.ndim <= 2 && sub != 1) {
        // more than two ghost layers: the sweep kernels copy through the two layers their strips hold; the outer
        // layers of qnew are copies of qold as well (step2ds.f / step1.f update interior cells of a copy).
        // Only the layers no tile of this pass writes are copied (x pass: outer ghost columns of every row, y pass: outer
        // ghost rows), with the pass' own boundary conditions where it evaluates them while loading (a.vbc_on: those
        // ghost cells were never filled in memory).  In an overlapped decomposed step the copy goes with the RIM
        // launch (sub == 2), i.e. behind the halo exchange on its stream: the interior launch (sub == 1) runs while the
        // unpack writes those very ghost cells of qin.
        FrameBc f;
        for (int k = 0; k < 4; k++) {
            const int t = a.vbc_on ? a.vbc[k] : -1;
            f.t[k] = t == PCL_BC_CUSTOM ? 100 : t;
            for (int m = 0; m < 8; m++) f.c[k][m] = a.vbc_on ? a.vconst[k][m] : 0.0;
        }
        f.only = ids;
        hipLaunchKernelGGL(frame_kernel, dim3(frame_blocks(s)), dim3(256), 0, stream, const_cast<double *>(qin), qout,
                           s->cfg.meqn, s->I, s->J, s->cfg.mbc, s->pitch, s->plane, f);
        HIP_TRY(hipGetLastError());
    }
    pcl_solver::Timed t{};
    const bool timed = timing_on(s) && !on;  // a launch on the halo stream runs beside the interior: not timed
    if (timed) {
        t.a = get_event(s);
        t.b = get_event(s);
        t.which = ids - 1;
        t.count = sub != 2;
        hipEventRecord(t.a, stream);
    }
    SweepLaunch l;
    l.a = a;
    l.ndim = s->cfg.ndim;
    l.rp = s->cfg.rp;
    l.ids = ids;
    l.fwave = s->cfg.fwave;
    l.stream = stream;
    std::string err;
    int rc = PCL_BY_MATH(s->cfg.math, launch_sweep(l, err));
    if (rc) fail(rc, err);
    if (timed) {
        hipEventRecord(t.b, stream);
        s->timed.push_back(t);
        if (s->timed.size() >= 2048) drain_timing(s);
    }
    return rc;
}

// The dimension-split 2-D step as ONE kernel (classic_fused.hpp): two ghost layers, Riemann solvers without aux
// arrays, no capacity function.  PCL_TUNE_FUSED_STEP=0 keeps the two passes.
// sub/box: tile subset (interior / rim of a decomposed block), `on`: the halo stream for the rim launch.
int fused_step_mode() {
    static const int on = [] { const char *e = getenv("PCL_TUNE_FUSED_STEP"); return e ? atoi(e) : 2; }();
    return on;
}
bool fused_step_ok(const pcl_solver *s) {
    const int on = fused_step_mode();
    const int rp = s->cfg.rp;
    return on && s->cfg.ndim == 2 && s->cfg.method[2] < 0 && s->cfg.mbc == 2 && s->cfg.method[5] <= 0 && s->cfg.meqn <= 5 &&
           (rp == PCL_RP_EULER5_2D || rp == PCL_RP_ACOUSTICS_2D || rp == PCL_RP_ADVECTION_2D || rp == PCL_RP_SHALLOW_2D);
}
// The two-pass dim-split step of a decomposed block with its interior x tiles BESIDE the exchange.  For the solver
// family of the one-kernel step (aux-free, no capacity function: an x pass register-allocated for four workgroups per
// CU) the interior launch starves the halo stream's pack / Send-Recv kernels (44 / 77 us instead of 6 / 34) and the step
// is slower than with the exchange in front (4096 x 2048 Euler block: 0.489 against 0.320 ms), so those blocks -- too
// thin for one-kernel tiles, mbc > 2, PCL_TUNE_FUSED_STEP=0 -- take the exchange in front unless PCL_HALO_OVERLAP=1 is
// set explicitly (tests, A/B).  Solvers with aux arrays or a capacity function keep the overlap.
bool twopass_overlap_ok(const pcl_solver *s) {
    const int rp = s->cfg.rp;
    const bool onek_family = s->cfg.ndim == 2 && s->cfg.method[2] < 0 && s->cfg.method[5] <= 0 &&
                             (rp == PCL_RP_EULER5_2D || rp == PCL_RP_ACOUSTICS_2D || rp == PCL_RP_ADVECTION_2D || rp == PCL_RP_SHALLOW_2D);
    return !(onek_family && s->overlap_dflt);
}
int do_step2ds(pcl_solver *s, const double *qin, double *qout, double dt, int sub = 0, const int *box = nullptr,
               hipStream_t on = nullptr) {
    hipStream_t stream = on ? on : s->stream;
    SweepArgs a = make_args(s, qin, qout, 1, dt);
    a.dtd_t = dt / s->cfg.d[1];
    a.src_id = s->fused_src;
    a.sub = sub;
    if (sub) for (int k = 0; k < 4; k++) a.box[k] = box[k];
    pcl_solver::Timed t{};
    const bool timed = timing_on(s) && !on;
    if (timed) {
        t.a = get_event(s);
        t.b = get_event(s);
        t.which = 2;
        t.count = sub != 2;
        hipEventRecord(t.a, stream);
    }
    SweepLaunch l;
    l.a = a;
    l.ndim = 2;
    l.rp = s->cfg.rp;
    l.ids = 1;
    l.fwave = s->cfg.fwave;
    l.stream = stream;
    std::string err;
    int rc = PCL_BY_MATH(s->cfg.math, launch_step2ds(l, err));
    if (rc) fail(rc, err);
    if (timed) {
        hipEventRecord(t.b, stream);
        s->timed.push_back(t);
        if (s->timed.size() >= 2048) drain_timing(s);
    }
    return rc;
}

// 3-D dimension-split sweep along dir (1..3), qin -> qout (step3ds.f; kernel in classic.hpp)
int do_sweep3(pcl_solver *s, const double *qin, double *qout, int dir, double dt) {
    SweepArgs a = make_args(s, qin, qout, 1, dt);
    const int mbc = s->cfg.mbc;
    const long st[3] = {1, s->pitch, s->pitch * s->J};
    const int n[3] = {s->I, s->J, s->K};
    // (along, across, batch): x: (i, j, k)   y: (j, i, k)   z: (k, i, j) -- "across" is i whenever it can be
    const int al = dir - 1, ac = dir == 1 ? 1 : 0, bt = dir == 3 ? 1 : 2;
    a.dtd = dt / s->cfg.d[al];
    a.dx = s->cfg.d[al];
    a.s_al = st[al]; a.s_ac = st[ac]; a.s_b = st[bt];
    a.n_al = n[al]; a.n_ac = n[ac]; a.n_b = n[bt];
    a.m_al = s->cfg.n[al];
    a.lo_ac = mbc - 1; a.hi_ac = mbc + s->cfg.n[ac];   // slices 0..m+1: one ghost layer (step3ds.f:110-111)
    a.lo_b = mbc - 1; a.hi_b = mbc + s->cfg.n[bt];
    a.vbc_on = 0;
    pcl_solver::Timed t{};
    if (timing_on(s)) {
        t.a = get_event(s);
        t.b = get_event(s);
        t.which = dir == 1 ? 0 : 1;
        t.count = true;
        hipEventRecord(t.a, s->stream);
    }
    SweepLaunch l;
    l.a = a;
    l.ndim = 3;
    l.rp = s->cfg.rp;
    l.ids = dir;
    l.fwave = s->cfg.fwave;
    l.stream = s->stream;
    std::string err;
    int rc = PCL_BY_MATH(s->cfg.math, launch_sweep3(l, err));
    if (rc) fail(rc, err);
    if (timing_on(s)) {
        hipEventRecord(t.b, s->stream);
        s->timed.push_back(t);
        if (s->timed.size() >= 2048) drain_timing(s);
    }
    return rc;
}

// unsplit 3-D step (step3.f): for x, y, z in turn, every slice's pieces into scratch planes and the ordered combine into
// t1 (kernels in classic3.hpp).  All three directions read the same qold = q.
int do_unsplit3(pcl_solver *s, double dt) {
    const int m3 = s->cfg.method[2] / 10, m4 = s->cfg.method[2] - 10 * m3;
    if (m3 < 0 || m3 > 2 || m4 < 0 || m4 > 2 || (m4 > 0 && m3 == 0))
        return fail(PCL_EINVAL, "3-D order_trans must be 0, 10, 11, 20, 21 or 22 (flux3.f:46-73)");
    const size_t qbytes = ((size_t)s->total + 16) * sizeof(double);
    // the marching kernels (classic3.hpp) keep the slices' pieces in registers and LDS; only the scratch-plane form
    // (PCL_TUNE_UNSPLIT3=0, kept for A/B runs) needs the 14 plane sets
    static const int marching = [] { const char *e = getenv("PCL_TUNE_UNSPLIT3"); return e ? atoi(e) : 1; }();
    for (int k = 0; k < 14 && !marching; k++) {
        if (s->scr3[k]) continue;
        double *raw = nullptr;
        HIP_TRY(hipMalloc((void **)&raw, qbytes));
        HIP_TRY(hipMemsetAsync(raw, 0, qbytes, s->stream));
        s->scr3[k] = raw + s->lead;
    }
    const int mbc = s->cfg.mbc;
    const long st[3] = {1, s->pitch, s->pitch * s->J};
    const int n[3] = {s->I, s->J, s->K};
    std::string err;
    for (int dir = 1; dir <= 3; dir++) {
        const int d = dir - 1, e = (d + 1) % 3, f = (d + 2) % 3;
        Unsplit3Launch l;
        l.a = make_args(s, s->q, s->t1, 1, dt);
        l.a.dtd = dt / s->cfg.d[d];
        l.a.dx = s->cfg.d[d];
        l.a.s_al = st[d]; l.a.n_al = n[d]; l.a.m_al = s->cfg.n[d];
        l.a.vbc_on = 0;
        for (int k = 0; k < 14; k++) l.scr[k] = s->scr3[k];
        l.qacc = s->t1;
        l.s_e = st[e]; l.s_f = st[f]; l.n_e = n[e]; l.n_f = n[f]; l.m_e = s->cfg.n[e]; l.m_f = s->cfg.n[f];
        l.m3 = m3; l.m4 = m4;
        l.dty = dt / s->cfg.d[e]; l.dtz = dt / s->cfg.d[f];
        l.dir = dir; l.rp = s->cfg.rp; l.stream = s->stream;
        (void)mbc;
        pcl_solver::Timed t{};
        if (timing_on(s)) { t.a = get_event(s); t.b = get_event(s); t.which = dir == 1 ? 0 : 1; t.count = true; hipEventRecord(t.a, s->stream); }
        int rc = PCL_BY_MATH(s->cfg.math, launch_unsplit3(l, err));
        if (timing_on(s)) { hipEventRecord(t.b, s->stream); s->timed.push_back(t); }
        if (rc) return fail(rc, err);
    }
    return PCL_OK;
}

static int unsplit_frame(pcl_solver *s, hipStream_t stream, const int *bc = nullptr, const double *cstate = nullptr) {
    FrameBc f;
    f.only = 0;
    for (int k = 0; k < 4; k++) {
        f.t[k] = bc ? (bc[k] == PCL_BC_CUSTOM ? 100 : bc[k]) : -1;
        for (int m = 0; m < 8; m++) f.c[k][m] = (bc && bc[k] == PCL_BC_CUSTOM && cstate) ? cstate[k * PCL_MAX_RP_PARAMS + m] : 0.0;
    }
    hipLaunchKernelGGL(frame_kernel, dim3(frame_blocks(s)), dim3(256), 0, stream, s->q, s->t1, s->cfg.meqn, s->I, s->J,
                       s->cfg.mbc, s->pitch, s->plane, f);
    HIP_TRY(hipGetLastError());
    return PCL_OK;
}

// one phase (ids = 1: x, 2: y); sub = 0 all tiles, 1 the tiles that read no ghost cell, 2 the others (x phase only)
static int unsplit_phase(pcl_solver *s, int ids, double dt, int sub, hipStream_t stream) {
    std::string err;
    SweepLaunch l;
    l.a = make_args(s, s->q, s->t1, ids, dt);
    l.a.trans = s->cfg.method[2];
    l.a.dtd_t = dt / s->cfg.d[2 - ids];
    l.a.sub = sub;
    l.ndim = 2; l.rp = s->cfg.rp; l.ids = ids; l.fwave = s->cfg.fwave; l.stream = stream;
    pcl_solver::Timed t{};
    if (timing_on(s)) { t.a = get_event(s); t.b = get_event(s); t.which = ids - 1; t.count = sub != 2; hipEventRecord(t.a, stream); }
    int rc = PCL_BY_MATH(s->cfg.math, launch_unsplit(l, s->t1, err));
    if (timing_on(s)) { hipEventRecord(t.b, stream); s->timed.push_back(t); }
    if (rc) return fail(rc, err);
    return PCL_OK;
}

int do_unsplit_lds(pcl_solver *s, double dt) {
    if (int rc = unsplit_frame(s, s->stream)) return rc;
    for (int ids = 1; ids <= 2; ids++)
        if (int rc = unsplit_phase(s, ids, dt, 0, s->stream)) return rc;
    return PCL_OK;
}

int do_unsplit(pcl_solver *s, double dt) { return do_unsplit_lds(s, dt); }

// a failed step must not leave a partial maximum behind (see read_cfl's invariant)
int bail(pcl_solver *s, int rc) {
    (void)hipMemsetAsync(s->cfl_dev, 0, sizeof(unsigned long long), s->stream);
    return rc;
}

__global__ void cfl_handover(unsigned long long *word, unsigned long long *host, unsigned long long seq) {
    const unsigned long long v = *word;
    *word = 0;                       // invariant: the word is zero whenever no step is in flight
    __hip_atomic_store(host, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(host + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The step's Courant number: in a decomposed run the max over all blocks (petclaw/cfl.py:29-31),
// reduced on the device before the single 8-byte read-back.
int read_cfl_begin(pcl_solver *s) {
    s->step_no++;           // every launch of the step / stage is enqueued: the next one decides afresh whether it is timed
    if (s->halo.active) {
        std::string err;
        if (s->halo.allreduce_max_device(reinterpret_cast<double *>(s->cfl_dev), err)) return fail(PCL_ECOMM, err);
    }
    return PCL_OK;
}
int read_cfl_end(pcl_solver *s, double *cfl);
int read_cfl(pcl_solver *s, double *cfl) {
    if (int rc = read_cfl_begin(s)) return rc;
    return read_cfl_end(s, cfl);
}
int read_cfl_end(pcl_solver *s, double *cfl) {
    if (s->cfl_poll) {
        // One single-thread kernel behind the sweeps hands the word over through host memory the device can
        // write (the value, then a sequence number with release semantics at system scope) and re-zeroes it;
        // the host polls the sequence number.  ~3 us per step less than a D2H copy + event wait.
        const unsigned long long seq = ++s->cfl_seq;
        hipLaunchKernelGGL(cfl_handover, dim3(1), dim3(1), 0, s->stream, s->cfl_dev, s->cfl_host_dev, seq);
        HIP_TRY(hipGetLastError());
        unsigned long long *flag = s->cfl_host + 1;
        long spins = 0;
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
            if (++spins > 4000000) {   // a long while without an answer: let the runtime report what happened
                HIP_TRY(hipStreamSynchronize(s->stream));
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) return fail(PCL_EHIP, "CFL hand-over never arrived");
                break;
            }
            __builtin_ia32_pause();
        }
    } else {
        HIP_TRY(hipMemcpyAsync(s->cfl_host, s->cfl_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost,
                               s->stream));
        HIP_TRY(hipEventRecord(s->ev_cfl, s->stream));
        // Invariant: the CFL word is zero whenever no step is in flight.  Re-zeroing it here, behind the
        // read-back, keeps the reset off the critical path (the host only waits for the copy).
        HIP_TRY(hipMemsetAsync(s->cfl_dev, 0, sizeof(unsigned long long), s->stream));
        HIP_TRY(hipEventSynchronize(s->ev_cfl));
    }
    double v;
    memcpy(&v, s->cfl_host, sizeof(double));
    *cfl = v;
    return PCL_OK;
}

int check_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(PCL_ENODEVICE, "no HIP device available (libpyclaw_amd has no CPU path)");
    return PCL_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

const char *pcl_last_error(void) { return g_err.c_str(); }
int pcl_version(void) { return 100; }
int pcl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pcl_create(const pcl_config *cfg, pcl_solver **out) {
    if (!cfg || !out) return fail(PCL_EINVAL, "null argument");
    if (getenv("PCL_TRACE_ABORT")) {   // diagnostics: C backtrace on abort()
        signal(SIGABRT, [](int) {
            void *bt[64];
            const int n = backtrace(bt, 64);
            backtrace_symbols_fd(bt, n, 2);
            signal(SIGABRT, SIG_DFL);
            raise(SIGABRT);
        });
    }
    *out = nullptr;
    if (cfg->ndim < 1 || cfg->ndim > 3) return fail(PCL_EINVAL, "ndim must be 1, 2 or 3");
    // the reference's default is 2 (the limiter reaches two cells); more ghost layers only widen the frame that the
    // sweeps copy through / sweep over (step2ds.f sweeps every ghost row), the strips keep their 2-cell halo
    if (cfg->kind == PCL_KIND_CLASSIC && (cfg->mbc < 2 || cfg->mbc > 8))
        return fail(PCL_EINVAL, "classic kernels need 2 <= mbc <= 8");
    if (cfg->kind == PCL_KIND_CLASSIC && cfg->ndim == 3 && cfg->mbc != 2)
        return fail(PCL_EINVAL, "3-D classic kernels need mbc == 2 (reference default)");
    // mbc = (weno_order+1)/2 (sharpclaw.py:479): 3 for tvd2 / WENO5 / legacy WENO5, 4..9 = weno_order 7..17 (lim_type 2)
    if (cfg->kind == PCL_KIND_SHARPCLAW && (cfg->mbc < 3 || cfg->mbc > 9))
        return fail(PCL_EINVAL, "SharpClaw: mbc = (weno_order+1)/2 must be 3..9 (weno_order 5..17)");
    if (cfg->kind == PCL_KIND_SHARPCLAW && cfg->mbc > 3 && cfg->lim_type != 2)
        return fail(PCL_EINVAL, "SharpClaw: weno_order > 5 (mbc > 3) exists for lim_type 2 only");
    if (cfg->kind != PCL_KIND_CLASSIC && cfg->kind != PCL_KIND_SHARPCLAW) return fail(PCL_EINVAL, "unknown solver kind");
    if (cfg->kind == PCL_KIND_SHARPCLAW && cfg->lim_type != 1 && cfg->lim_type != 2 && cfg->lim_type != 3)
        return fail(PCL_EINVAL, "SharpClaw: lim_type must be 1 (tvd2), 2 (WENO5) or 3 (legacy WENO5)");
    if (cfg->kind == PCL_KIND_SHARPCLAW && cfg->method[4] != 0) {
        // char_decomp (sharpclaw.py:262): 1 = wave-based reconstruction, 1-D only (the reference's 2-D flux1.f90 calls rpn2
        // with a wrong argument list on that path, 2d/sharpclaw/flux1.f90:86); 2 and 3 need a user-supplied evec routine
        // the reference only stubs (evec.f90:13-14)
        if (cfg->method[4] != 1) return fail(PCL_EINVAL, "SharpClaw: char_decomp must be 0 or 1 (2, 3 need a user evec routine)");
        if (cfg->ndim != 1 || cfg->mbc != 3 || (cfg->lim_type != 1 && cfg->lim_type != 2) || cfg->method[5] != 0 || cfg->fwave)
            return fail(PCL_EINVAL, "SharpClaw char_decomp = 1: 1-D, lim_type 1 (tvd2_wave) or 2 (weno5_wave), weno_order 5, "
                                    "no capacity function, no f-wave solver");
    }
    if (cfg->kind == PCL_KIND_SHARPCLAW && cfg->lim_type == 1 && cfg->meqn > PCL_MAX_WAVES)
        return fail(PCL_EINVAL, "SharpClaw tvd2: mthlim is indexed by component, meqn <= PCL_MAX_WAVES");
    if (cfg->mwaves < 1 || cfg->mwaves > PCL_MAX_WAVES) return fail(PCL_EINVAL, "bad mwaves");
    if (cfg->math != PCL_MATH_EXACT && cfg->math != PCL_MATH_FAST && cfg->math != PCL_MATH_STRICT)
        return fail(PCL_EINVAL, "unknown math mode");
    int want_meqn = 0, want_mwaves = 0, want_ndim = 0;
    switch (cfg->rp) {
    case PCL_RP_ADVECTION_1D: want_meqn = 1; want_mwaves = 1; want_ndim = 1; break;
    case PCL_RP_ACOUSTICS_1D: want_meqn = 2; want_mwaves = 2; want_ndim = 1; break;
    case PCL_RP_BURGERS_1D: want_meqn = 1; want_mwaves = 1; want_ndim = 1; break;
    case PCL_RP_EULER_1D: want_meqn = 3; want_mwaves = 3; want_ndim = 1; break;
    case PCL_RP_SHALLOW_1D: want_meqn = 2; want_mwaves = 2; want_ndim = 1; break;
    case PCL_RP_ADVECTION_COLOR_1D: want_meqn = 1; want_mwaves = 1; want_ndim = 1; break;
    case PCL_RP_ELASTICITY_FWAVE_1D: want_meqn = 2; want_mwaves = 2; want_ndim = 1; break;
    case PCL_RP_PSYSTEM_FWAVE_2D: want_meqn = 3; want_mwaves = 2; want_ndim = 2; break;
    case PCL_RP_ADVECTION_2D: want_meqn = 1; want_mwaves = 1; want_ndim = 2; break;
    case PCL_RP_SHALLOW_2D: want_meqn = 3; want_mwaves = 3; want_ndim = 2; break;
    case PCL_RP_VC_ACOUSTICS_2D: want_meqn = 3; want_mwaves = 2; want_ndim = 2; break;
    case PCL_RP_VC_ADVECTION_2D: want_meqn = 1; want_mwaves = 1; want_ndim = 2; break;
    case PCL_RP_SHALLOW_SPHERE_2D: want_meqn = 4; want_mwaves = 3; want_ndim = 2; break;
    case PCL_RP_ACOUSTICS_2D: want_meqn = 3; want_mwaves = 2; want_ndim = 2; break;
    case PCL_RP_EULER5_2D: want_meqn = 5; want_mwaves = 5; want_ndim = 2; break;
    case PCL_RP_VC_ACOUSTICS_3D: want_meqn = 4; want_mwaves = 2; want_ndim = 3; break;
    default: return fail(PCL_EINVAL, "unknown Riemann solver id");
    }
    if (cfg->meqn != want_meqn || cfg->mwaves != want_mwaves || cfg->ndim != want_ndim)
        return fail(PCL_EINVAL, "meqn/mwaves/ndim do not match the Riemann solver");
    for (int d = 0; d < cfg->ndim; d++)
        if (cfg->n[d] < 1) return fail(PCL_EINVAL, "grid extent must be >= 1");
    if (cfg->method[5] < 0 || cfg->method[5] > cfg->maux) return fail(PCL_EINVAL, "mcapa out of range");
    if (cfg->rp == PCL_RP_ADVECTION_COLOR_1D && cfg->maux < 1)
        return fail(PCL_EINVAL, "rp1_advection_color needs aux(1) = edge velocity");
    if (cfg->rp == PCL_RP_VC_ACOUSTICS_2D || cfg->rp == PCL_RP_VC_ADVECTION_2D) {
        if (cfg->maux < 2) return fail(PCL_EINVAL, "this Riemann solver needs two aux components (impedance/sound speed or the edge velocities)");
    }
    if (cfg->rp == PCL_RP_ELASTICITY_FWAVE_1D || cfg->rp == PCL_RP_PSYSTEM_FWAVE_2D) {
        if (!cfg->fwave) return fail(PCL_EINVAL, "this Riemann solver returns f-waves: cfg.fwave must be 1 (classic1fw / classic2fw)");
        if (cfg->maux < 3) return fail(PCL_EINVAL, "f-wave elasticity solvers need aux(1)=rho, aux(2)=K, aux(3)=stress-law flag");
        if (cfg->rp == PCL_RP_PSYSTEM_FWAVE_2D && cfg->method[2] > 0 && cfg->maux < 4)
            return fail(PCL_EINVAL, "rpt2_psystem reads aux(4) = strain of the neighbouring rows (the app's b4step fills it)");
        if (cfg->kind == PCL_KIND_SHARPCLAW) return fail(PCL_EINVAL, "f-wave solvers are wired for the classic solvers only");
    } else if (cfg->fwave) {
        return fail(PCL_EINVAL, "cfg.fwave = 1 needs an f-wave Riemann solver (PCL_RP_ELASTICITY_FWAVE_1D, PCL_RP_PSYSTEM_FWAVE_2D)");
    }
    if (cfg->rp == PCL_RP_SHALLOW_SPHERE_2D) {
        if (cfg->maux < 16) return fail(PCL_EINVAL, "rpn2_shallow_sphere needs the 16 aux components of setaux.f (kappa, edge normals/tangents, radial vector)");
        if (!(cfg->rp_params[0] > 0.0) || !(cfg->rp_params[1] > 0.0) || !(cfg->rp_params[2] > 0.0))
            return fail(PCL_EINVAL, "rpn2_shallow_sphere: rp_params must be g, dxcom, dycom (all > 0)");
        if (cfg->kind == PCL_KIND_CLASSIC && cfg->method[2] >= 0 && cfg->method[5] == 0)
            return fail(PCL_EINVAL, "shallow water on the sphere, unsplit: the capacity function (mcapa) must be set (step2qcor.f)");
    }
    if (cfg->ndim == 3) {
        if (cfg->kind != PCL_KIND_CLASSIC) return fail(PCL_EINVAL, "3-D: classic solver only (the reference has no 3-D SharpClaw)");
        if (cfg->maux < 2) return fail(PCL_EINVAL, "rpn3_vc_acoustics needs aux(1)=impedance, aux(2)=sound speed");
        if (cfg->method[5] != 0 && cfg->method[2] >= 0)
            return fail(PCL_EINVAL, "3-D: the capacity function is built for the dimension-split step (step3ds); the unsplit step3 "
                                    "with mcapa is not");
    }
    if (int rc = check_device()) return rc;
    HIP_TRY(hipSetDevice(cfg->device));

    pcl_solver *s = new pcl_solver();
    s->cfg = *cfg;
    s->I = cfg->n[0] + 2 * cfg->mbc;
    s->J = cfg->ndim > 1 ? cfg->n[1] + 2 * cfg->mbc : 1;
    s->K = cfg->ndim > 2 ? cfg->n[2] + 2 * cfg->mbc : 1;
    s->pitch = ((long)s->I + 15) / 16 * 16;
    s->plane = s->pitch * s->J * s->K;
    s->total = s->plane * cfg->meqn;
    const size_t qbytes = (size_t)s->total * sizeof(double);
    // every buffer is zero-filled ON THE SOLVER'S STREAM: a null-stream hipMemset is not
    // ordered against this non-blocking stream and could land after the first upload
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    s->lead = 16 - cfg->mbc;
    auto alloc = [&](double **p, size_t bytes) {
        if (e != hipSuccess) return;
        double *raw = nullptr;
        bytes += 16 * sizeof(double);
        e = hipMalloc((void **)&raw, bytes);
        if (e == hipSuccess) e = hipMemsetAsync(raw, 0, bytes, s->stream);
        if (e == hipSuccess) *p = raw + s->lead;
    };
    alloc(&s->q, qbytes);
    if (cfg->kind == PCL_KIND_CLASSIC) {
        alloc(&s->t1, qbytes);
        if (cfg->ndim > 1) alloc(&s->t2, qbytes);
        if (cfg->ndim > 2) alloc(&s->t3, qbytes);
    } else {
        for (int k = 1; k < 5; k++) alloc(&s->sreg[k], qbytes);
    }
    if (cfg->maux > 0) alloc(&s->aux, (size_t)s->plane * cfg->maux * sizeof(double));
    const int nmax = cfg->meqn > cfg->maux ? cfg->meqn : cfg->maux;
    s->stage_bytes = (size_t)nmax * s->I * s->J * s->K * sizeof(double);
    alloc(&s->stage, s->stage_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&s->cfl_dev, 64);
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->cfl_host, 64, hipHostMallocDefault);
    if (e == hipSuccess) { memset(s->cfl_host, 0, 64); e = hipHostGetDevicePointer((void **)&s->cfl_host_dev, s->cfl_host, 0); }
    { const char *p = getenv("PCL_CFL_POLL"); s->cfl_poll = p ? atoi(p) : 1; }
    if (e == hipSuccess) e = hipEventCreate(&s->ev0);
    if (e == hipSuccess) e = hipEventCreate(&s->ev1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_cfl, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMemsetAsync(s->cfl_dev, 0, 64, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) {
        std::string msg = std::string("pcl_create: ") + hipGetErrorString(e);
        pcl_destroy(s);
        return fail(PCL_EHIP, msg);
    }
    *out = s;
    return PCL_OK;
}

void pcl_destroy(pcl_solver *s) {
    if (!s) return;
    hipSetDevice(s->cfg.device);
    if (s->stream) hipStreamSynchronize(s->stream);
    if (s->hstream) hipStreamSynchronize(s->hstream);
    s->halo.destroy();
    if (s->ev_h0) hipEventDestroy(s->ev_h0);
    if (s->ev_h1) hipEventDestroy(s->ev_h1);
    if (s->ev_y) hipEventDestroy(s->ev_y);
    if (s->hstream) hipStreamDestroy(s->hstream);
    for (auto &t : s->timed) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    for (auto &e : s->evpool) hipEventDestroy(e);
    for (double *p : {s->q, s->t1, s->t2, s->t3, s->bak, s->aux, s->stage})
        if (p) hipFree(p - s->lead);
    for (double *p : s->scr3)
        if (p) hipFree(p - s->lead);
    for (int k = 1; k < 5; k++)
        if (s->sreg[k]) hipFree(s->sreg[k] - s->lead);
    hipFree(s->cfl_dev);
    if (s->cfl_host) hipHostFree(s->cfl_host);
    if (s->ev0) hipEventDestroy(s->ev0);
    if (s->ev1) hipEventDestroy(s->ev1);
    if (s->ev_cfl) hipEventDestroy(s->ev_cfl);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
}

static int put_array(pcl_solver *s, const double *host, double *dev, int nm, int with_ghosts) {
    const int mbc = s->cfg.mbc;
    const int ni = with_ghosts ? s->I : s->cfg.n[0];
    const int nj = s->cfg.ndim > 1 ? (with_ghosts ? s->J : s->cfg.n[1]) : 1;
    const int io = with_ghosts ? 0 : mbc;
    const int jo = (s->cfg.ndim > 1 && !with_ghosts) ? mbc : 0;
    const int nk = s->cfg.ndim > 2 ? (with_ghosts ? s->K : s->cfg.n[2]) : 1;
    const int ko = (s->cfg.ndim > 2 && !with_ghosts) ? mbc : 0;
    const size_t bytes = (size_t)nm * ni * nj * nk * sizeof(double);
    HIP_TRY(hipMemcpyAsync(s->stage, host, bytes, hipMemcpyHostToDevice, s->stream));
    dim3 grid((ni + 255) / 256, nj, nk);
    hipLaunchKernelGGL(aos_to_soa, grid, dim3(256), 0, s->stream, s->stage, dev, nm, ni, nj, io, jo,
                       s->pitch, s->plane, ko, s->pitch * s->J);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    return PCL_OK;
}

int pcl_put_q(pcl_solver *s, const double *host, int with_ghosts) {
    if (s) s->ghosts_drop_all();          // exchange-ahead: whatever filled the ghost frames no longer holds
    if (!s || !host) return fail(PCL_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    s->undo_slot = nullptr;
    return put_array(s, host, cur(s), s->cfg.meqn, with_ghosts);
}

int pcl_put_aux(pcl_solver *s, const double *host) {
    if (!s || !host) return fail(PCL_EINVAL, "null argument");
    if (s->cfg.maux <= 0) return fail(PCL_EINVAL, "solver was created with maux == 0");
    HIP_TRY(hipSetDevice(s->cfg.device));
    return put_array(s, host, s->aux, s->cfg.maux, 1);
}

int pcl_get_q(pcl_solver *s, double *host, int with_ghosts) {
    if (!s || !host) return fail(PCL_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    const int mbc = s->cfg.mbc, nm = s->cfg.meqn;
    const int ni = with_ghosts ? s->I : s->cfg.n[0];
    const int nj = s->cfg.ndim > 1 ? (with_ghosts ? s->J : s->cfg.n[1]) : 1;
    const int io = with_ghosts ? 0 : mbc;
    const int jo = (s->cfg.ndim > 1 && !with_ghosts) ? mbc : 0;
    const int nk = s->cfg.ndim > 2 ? (with_ghosts ? s->K : s->cfg.n[2]) : 1;
    const int ko = (s->cfg.ndim > 2 && !with_ghosts) ? mbc : 0;
    const size_t bytes = (size_t)nm * ni * nj * nk * sizeof(double);
    dim3 grid((ni + 255) / 256, nj, nk);
    hipLaunchKernelGGL(soa_to_aos, grid, dim3(256), 0, s->stream, cur(s), s->stage, nm, ni, nj, io, jo,
                       s->pitch, s->plane, ko, s->pitch * s->J);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host, s->stage, bytes, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return PCL_OK;
}

// window of a ghost strip: `width` layers of side `side` of dimension idim, the whole ghosted extent in the others
struct StripWin { int ni, nj, nk, io, jo, ko; };
static int strip_window(pcl_solver *s, int idim, int side, int width, StripWin &w) {
    if (idim < 0 || idim >= s->cfg.ndim) return fail(PCL_EINVAL, "bad idim");
    const int N = idim == 0 ? s->I : (idim == 1 ? s->J : s->K);
    if (width < 1 || width > N) return fail(PCL_EINVAL, "bad strip width");
    w = StripWin{s->I, s->J, s->K, 0, 0, 0};
    if (idim == 0) { w.ni = width; w.io = side == 0 ? 0 : N - width; }
    else if (idim == 1) { w.nj = width; w.jo = side == 0 ? 0 : N - width; }
    else { w.nk = width; w.ko = side == 0 ? 0 : N - width; }
    return PCL_OK;
}

int pcl_get_strip(pcl_solver *s, int idim, int side, int width, double *host) {
    if (!s || !host) return fail(PCL_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    StripWin w;
    if (int rc = strip_window(s, idim, side, width, w)) return rc;
    const int nm = s->cfg.meqn;
    dim3 grid((w.ni + 255) / 256, w.nj, w.nk);
    hipLaunchKernelGGL(soa_to_aos, grid, dim3(256), 0, s->stream, cur(s), s->stage, nm, w.ni, w.nj, w.io, w.jo,
                       s->pitch, s->plane, w.ko, s->pitch * s->J);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host, s->stage, (size_t)nm * w.ni * w.nj * w.nk * sizeof(double), hipMemcpyDeviceToHost,
                           s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return PCL_OK;
}

int pcl_put_strip(pcl_solver *s, int idim, int side, int width, const double *host) {
    if (s) s->ghosts_drop_all();          // exchange-ahead: whatever filled the ghost frames no longer holds
    if (!s || !host) return fail(PCL_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    StripWin w;
    if (int rc = strip_window(s, idim, side, width, w)) return rc;
    const int nm = s->cfg.meqn;
    HIP_TRY(hipMemcpyAsync(s->stage, host, (size_t)nm * w.ni * w.nj * w.nk * sizeof(double), hipMemcpyHostToDevice,
                           s->stream));
    dim3 grid((w.ni + 255) / 256, w.nj, w.nk);
    hipLaunchKernelGGL(aos_to_soa, grid, dim3(256), 0, s->stream, s->stage, cur(s), nm, w.ni, w.nj, w.io, w.jo,
                       s->pitch, s->plane, w.ko, s->pitch * s->J);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    return PCL_OK;
}

// the same for the aux array: a ghost strip computed by a Python aux-BC callback on the host (decomposed runs)
int pcl_put_aux_strip(pcl_solver *s, int idim, int side, int width, const double *host) {
    if (!s || !host) return fail(PCL_EINVAL, "null argument");
    if (s->cfg.maux <= 0 || !s->aux) return fail(PCL_ESTATE, "pcl_put_aux_strip: no aux array on the device");
    HIP_TRY(hipSetDevice(s->cfg.device));
    StripWin w;
    if (int rc = strip_window(s, idim, side, width, w)) return rc;
    const int nm = s->cfg.maux;
    const size_t bytes = (size_t)nm * w.ni * w.nj * w.nk * sizeof(double);
    if (bytes > s->stage_bytes) return fail(PCL_EINVAL, "pcl_put_aux_strip: strip exceeds the staging buffer");
    HIP_TRY(hipMemcpyAsync(s->stage, host, bytes, hipMemcpyHostToDevice, s->stream));
    dim3 grid((w.ni + 255) / 256, w.nj, w.nk);
    hipLaunchKernelGGL(aos_to_soa, grid, dim3(256), 0, s->stream, s->stage, s->aux, nm, w.ni, w.nj, w.io, w.jo, s->pitch, s->plane,
                       w.ko, s->pitch * s->J);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    return PCL_OK;
}

__global__ void gather_cells_kernel(const double *q, const double *aux, const int *ij, double *out, int ncell,
                                    int nq, int na, int mbc, int ndim, long pitch, long plane, long slab) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = nq + na;
    if (t >= ncell * per) return;
    const int c = t / per, m = t % per;
    const int st = ndim > 2 ? 3 : 2;                  // index tuples: (i, j) pairs in 1-D / 2-D, (i, j, k) triples in 3-D
    const long cell = (long)(ndim > 2 ? ij[st * c + 2] + mbc : 0) * slab + (long)(ndim > 1 ? ij[st * c + 1] + mbc : 0) * pitch +
                      ij[st * c] + mbc;
    out[t] = m < nq ? q[m * plane + cell] : aux[(m - nq) * plane + cell];
}

int pcl_get_cells(pcl_solver *s, int ncell, const int *ij, double *q, double *aux) {
    if (!s || !ij || !q) return fail(PCL_EINVAL, "null argument");
    if (ncell <= 0) return PCL_OK;
    const int st = s->cfg.ndim > 2 ? 3 : 2;
    const int nq = s->cfg.meqn, na = (aux && s->aux) ? s->cfg.maux : 0, per = nq + na;
    if (aux && s->cfg.maux > 0 && !s->aux) return fail(PCL_EINVAL, "pcl_get_cells: aux requested but never uploaded");
    for (int c = 0; c < ncell; c++) {
        const int i = ij[st * c], j = s->cfg.ndim > 1 ? ij[st * c + 1] : 0, k = s->cfg.ndim > 2 ? ij[st * c + 2] : 0;
        if (i < 0 || i >= s->cfg.n[0] || j < 0 || j >= (s->cfg.ndim > 1 ? s->cfg.n[1] : 1) ||
            k < 0 || k >= (s->cfg.ndim > 2 ? s->cfg.n[2] : 1))
            return fail(PCL_EINVAL, "pcl_get_cells: cell index outside the interior");
    }
    // staging layout: [ncell*per doubles of output][ncell*st ints of indices]
    const size_t out_bytes = (size_t)ncell * per * sizeof(double), idx_bytes = (size_t)ncell * st * sizeof(int);
    if (out_bytes + idx_bytes > s->stage_bytes) return fail(PCL_EINVAL, "pcl_get_cells: too many cells for the staging buffer");
    HIP_TRY(hipSetDevice(s->cfg.device));
    int *dij = reinterpret_cast<int *>(reinterpret_cast<char *>(s->stage) + out_bytes);
    HIP_TRY(hipMemcpyAsync(dij, ij, idx_bytes, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(gather_cells_kernel, dim3((ncell * per + 255) / 256), dim3(256), 0, s->stream, cur(s),
                       s->aux, dij, s->stage, ncell, nq, na, s->cfg.mbc, s->cfg.ndim, s->pitch, s->plane, s->pitch * s->J);
    HIP_TRY(hipGetLastError());
    std::vector<double> tmp((size_t)ncell * per);
    HIP_TRY(hipMemcpyAsync(tmp.data(), s->stage, out_bytes, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (int c = 0; c < ncell; c++) {
        for (int m = 0; m < nq; m++) q[(size_t)c * nq + m] = tmp[(size_t)c * per + m];
        for (int m = 0; m < na; m++) aux[(size_t)c * s->cfg.maux + m] = tmp[(size_t)c * per + nq + m];
    }
    return PCL_OK;
}

static int bc_launch(pcl_solver *s, int idim, int side, int type, const double *cstate, bool aux = false,
                     hipStream_t on = nullptr) {
    const hipStream_t stream = on ? on : s->stream;
    pcl::RpParams cs;  // the constant state travels as a kernel argument: no copy, no sync
    for (int k = 0; k < 8; k++) cs.v[k] = (cstate && k < s->cfg.meqn) ? cstate[k] : 0.0;
    const int nm = aux ? s->cfg.maux : s->cfg.meqn;
    if (s->cfg.ndim == 3) {
        if (type == 100) return fail(PCL_EINVAL, "constant-state BC is implemented for 1-D/2-D");
        Dims3 D{{s->I, s->J, s->K}, {1, s->pitch, s->pitch * s->J}};
        const int d1 = idim == 0 ? 1 : 0, d2 = idim == 2 ? 1 : 2;
        const long n3 = (long)D.n[d1] * D.n[d2] * s->cfg.mbc * nm;
        hipLaunchKernelGGL(bc3_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, stream,
                           aux ? s->aux : cur(s), nm, D, s->plane, s->cfg.mbc, idim, side, type, aux ? 1 : 0);
        HIP_TRY(hipGetLastError());
        return PCL_OK;
    }
    const int nt = idim == 0 ? s->J : s->I;
    const long n = (long)nt * s->cfg.mbc * nm;
    hipLaunchKernelGGL(bc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       aux ? s->aux : cur(s), nm, s->I, s->J, s->pitch, s->plane, s->cfg.mbc, idim, side, type,
                       cs, aux ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return PCL_OK;
}

int pcl_bc(pcl_solver *s, int idim, int side, int bctype) {
    if (s) s->ghosts_drop_all();          // exchange-ahead: whatever filled the ghost frames no longer holds
    if (!s) return fail(PCL_EINVAL, "null argument");
    if (idim < 0 || idim >= s->cfg.ndim || side < 0 || side > 1) return fail(PCL_EINVAL, "bad idim/side");
    HIP_TRY(hipSetDevice(s->cfg.device));
    if (bctype == PCL_BC_SPHERE_MIRROR) {
        if (s->cfg.ndim != 2 || idim != 1) return fail(PCL_EINVAL, "PCL_BC_SPHERE_MIRROR is the y boundary of a 2-D grid");
        const long n = (long)s->I * s->cfg.mbc * s->cfg.meqn;
        hipLaunchKernelGGL(bc_sphere_mirror, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, cur(s),
                           s->cfg.meqn, s->I, s->J, s->pitch, s->plane, s->cfg.mbc, side);
        HIP_TRY(hipGetLastError());
        return PCL_OK;
    }
    if (bctype != PCL_BC_OUTFLOW && bctype != PCL_BC_PERIODIC && bctype != PCL_BC_REFLECTING)
        return fail(PCL_EINVAL, "pcl_bc: only outflow/periodic/reflecting/sphere-mirror run here");
    return bc_launch(s, idim, side, bctype, nullptr);
}

int pcl_bc_aux(pcl_solver *s, int idim, int side, int bctype) {
    if (!s) return fail(PCL_EINVAL, "null argument");
    if (s->cfg.maux <= 0) return PCL_OK;
    if (idim < 0 || idim >= s->cfg.ndim || side < 0 || side > 1) return fail(PCL_EINVAL, "bad idim/side");
    if (bctype != PCL_BC_OUTFLOW && bctype != PCL_BC_PERIODIC && bctype != PCL_BC_REFLECTING)
        return fail(PCL_EINVAL, "pcl_bc_aux: only outflow/periodic/reflecting run here");
    HIP_TRY(hipSetDevice(s->cfg.device));
    return bc_launch(s, idim, side, bctype, nullptr, true);
}

int pcl_bc_const(pcl_solver *s, int idim, int side, const double *state) {
    if (s) s->ghosts_drop_all();          // exchange-ahead: whatever filled the ghost frames no longer holds
    if (!s || !state) return fail(PCL_EINVAL, "null argument");
    if (idim < 0 || idim >= s->cfg.ndim || side < 0 || side > 1) return fail(PCL_EINVAL, "bad idim/side");
    HIP_TRY(hipSetDevice(s->cfg.device));
    if (s->cfg.meqn > 8) return fail(PCL_EINVAL, "constant-state BC supports meqn <= 8");
    return bc_launch(s, idim, side, 100, state);
}

int pcl_sweep(pcl_solver *s, int ids, double dt, double *cfl) {
    if (s) s->ghosts_drop_all();          // exchange-ahead: whatever filled the ghost frames no longer holds
    if (!s || !cfl) return fail(PCL_EINVAL, "null argument");
    if (s->cfg.kind != PCL_KIND_CLASSIC) return fail(PCL_ESTATE, "classic call on a SharpClaw solver");
    if (ids < 1 || ids > s->cfg.ndim) return fail(PCL_EINVAL, "bad ids");
    HIP_TRY(hipSetDevice(s->cfg.device));
    if (int rc = s->cfg.ndim == 3 ? do_sweep3(s, s->q, s->t1, ids, dt) : do_sweep(s, s->q, s->t1, ids, dt))
        return bail(s, rc);
    std::swap(s->q, s->t1);
    s->undo_slot = &s->t1;
    return read_cfl(s, cfl);
}

int pcl_step_hyperbolic(pcl_solver *s, double dt, double *cfl) {
    if (s) s->ghosts_drop_all();          // exchange-ahead: whatever filled the ghost frames no longer holds
    if (!s || !cfl) return fail(PCL_EINVAL, "null argument");
    if (s->cfg.kind != PCL_KIND_CLASSIC) return fail(PCL_ESTATE, "classic call on a SharpClaw solver");
    HIP_TRY(hipSetDevice(s->cfg.device));
    if (s->cfg.ndim == 3) {  // Godunov splitting x, y, z (clawpack.py:674-690)
        if (s->cfg.method[2] >= 0) {     // unsplit, clawpack.py:690-696 -> step3.f
            if (int rc = do_unsplit3(s, dt)) return bail(s, rc);
            std::swap(s->q, s->t1);
            s->undo_slot = &s->t1;
            return read_cfl(s, cfl);
        }
        if (int rc = do_sweep3(s, s->q, s->t1, 1, dt)) return bail(s, rc);
        if (int rc = do_sweep3(s, s->t1, s->t2, 2, dt)) return bail(s, rc);
        if (int rc = do_sweep3(s, s->t2, s->t3, 3, dt)) return bail(s, rc);
        std::swap(s->q, s->t3);
        s->undo_slot = &s->t3;
    } else if (s->cfg.ndim == 1) {
        if (int rc = do_sweep(s, s->q, s->t1, 1, dt)) return bail(s, rc);
        std::swap(s->q, s->t1);
        s->undo_slot = &s->t1;
    } else if (s->cfg.method[2] < 0) {  // dimensional splitting, clawpack.py:538-546
        // one block: the faster form of the step, re-measured every FORM_WINDOW steps (pcl_solver::form_now)
        constexpr long FORM_WINDOW = 256;
        constexpr int FORM_TRIAL = 3;       // timed steps per form at the start of a window (the first one untimed)
        const bool can = fused_step_ok(s);
        // (a decomposed block whose exchange runs in front of the step, pcl_bc_step: every rank picks for itself -- both
        // forms see the same ghost frame, give the same bits and leave the order on the communicator alone)
        const bool tune = can && fused_step_mode() == 2 && (!s->halo.active || s->form_seq_tune);
        int form = can ? 1 : 0;
        constexpr long FORM_T0 = 64;        // the trials start 64 steps into a window: a short run never meets them
        if (tune) {
            const long k = s->form_step % FORM_WINDOW - FORM_T0;
            if (k == 0) s->form_t[0] = s->form_t[1] = 1e30;
            if (k >= 0 && k < 2 * (FORM_TRIAL + 1)) form = k < FORM_TRIAL + 1 ? 1 : 0;
            else {
                if (k == 2 * (FORM_TRIAL + 1)) s->form_now = s->form_t[0] < 0.98 * s->form_t[1] ? 0 : 1;
                form = s->form_now;
            }
        }
        const auto t0 = std::chrono::steady_clock::now();
        if (form == 1) {
            if (int rc = do_step2ds(s, s->q, s->t2, dt)) return bail(s, rc);
        } else {
            if (int rc = do_sweep(s, s->q, s->t1, 1, dt)) return bail(s, rc);
            if (int rc = do_sweep(s, s->t1, s->t2, 2, dt)) return bail(s, rc);
        }
        std::swap(s->q, s->t2);
        s->undo_slot = &s->t2;
        s->form_steps[form]++;
        const int rc = read_cfl(s, cfl);
        if (tune) {
            const long k = s->form_step % FORM_WINDOW - FORM_T0;
            if (k >= 0 && k < 2 * (FORM_TRIAL + 1) && k % (FORM_TRIAL + 1) != 0) {      // a trial step, not the first of its form
                const double dtw = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (dtw < s->form_t[form]) s->form_t[form] = dtw;
            }
            s->form_step++;
        }
        return rc;
    } else {  // unsplit, clawpack.py:550-552 -> step2.f
        if (int rc = do_unsplit(s, dt)) return bail(s, rc);
        std::swap(s->q, s->t1);
        s->undo_slot = &s->t1;
    }
    return read_cfl(s, cfl);
}

int pcl_bc_step(pcl_solver *s, const int *bc, const double *cstate, double dt, double *cfl) {
    if (!s || !bc || !cfl) return fail(PCL_EINVAL, "null argument");
    if (s->cfg.kind != PCL_KIND_CLASSIC) return fail(PCL_ESTATE, "classic call on a SharpClaw solver");
    HIP_TRY(hipSetDevice(s->cfg.device));
    for (int k = 0; k < 2 * s->cfg.ndim; k++) {
        const int t = bc[k];
        if (t == PCL_BC_SPHERE_MIRROR) {       // the sphere app's pole boundary: y sides of the unsplit 2-D step
            if (s->cfg.ndim != 2 || k < 2 || s->cfg.method[2] < 0)
                return fail(PCL_EINVAL, "PCL_BC_SPHERE_MIRROR in pcl_bc_step: y sides of the unsplit 2-D step only");
            continue;
        }
        if (t >= 0 && t != PCL_BC_CUSTOM && t != PCL_BC_OUTFLOW && t != PCL_BC_PERIODIC && t != PCL_BC_REFLECTING)
            return fail(PCL_EINVAL, "bad boundary condition type");
        if (t == PCL_BC_CUSTOM && !cstate) return fail(PCL_EINVAL, "constant state missing");
    }
    // Dimension-split steps (and 1-D): the first pass evaluates the boundary conditions while it
    // loads its tiles -- no ghost-fill launches (each costs ~5 us of launch latency per step).
    const bool fused = s->cfg.meqn <= 8 && (s->cfg.ndim == 1 || (s->cfg.ndim == 2 && s->cfg.method[2] < 0));
    // Decomposed dim-split 2-D step: the halo exchange runs on its own stream while the x pass does the
    // tiles that read no ghost cell; the rim tiles follow once the ghost frame has arrived.
    int box[4], ntiles[2];
    // both sweeps in one kernel (its own, taller tiles) where such tiles leave an interior; thin blocks keep the x pass'
    // 4-row tiles for the overlap
    bool onek = fused_step_ok(s), overlapped = false;
    if (s->halo.active && fused && s->cfg.ndim == 2 && s->overlap && s->sel == 0) {
        // (exchange-ahead in the two-pass order, agreed by the ranks because some block is too thin for one-kernel tiles:
        // this rank keeps that order too)
        if (s->exchange_ahead == 1) onek = false;
        if (onek && pcl::exact::step2ds_interior_box(make_args(s, s->q, s->t2, 1, dt), box, ntiles)) overlapped = true;
        else {
            onek = false;
            overlapped = (twopass_overlap_ok(s) || s->exchange_ahead == 1) &&
                         pcl::exact::x_interior_box(make_args(s, s->q, s->t1, 1, dt), box, ntiles);
        }
    }
    if (!overlapped) s->ghosts_drop_all();       // exchange-ahead lives in the overlapped dimension-split step only
    if (overlapped) {
        // a side without a neighbour block gets its ghost cells from the boundary conditions the kernel evaluates
        // while loading -- nothing there waits for the exchange, so the interior box reaches that edge
        if (!s->halo.has(pcl::S)) box[0] = 0;
        if (!s->halo.has(pcl::N)) box[1] = ntiles[0];
        if (!s->halo.has(pcl::W)) box[2] = 0;
        if (!s->halo.has(pcl::E)) box[3] = ntiles[1];
    }
    // Decomposed unsplit 2-D step: the ghost frame (exchange, then the physical BCs) is built on the halo stream while
    // the x phase runs the tiles that read no ghost cell; the rim tiles follow on the halo stream, the y phase (it reads
    // every cell of qold and of the x-phase result) after the join.  PCL_HALO_OVERLAP=2: the same launches on one
    // stream with the interior tiles strictly BEFORE the frame (race check, as for the dim-split step).  Without a
    // communicator the same code runs on one stream: ghost frame (one launch for all four sides), x phase, y phase.
    if (s->cfg.ndim == 2 && s->cfg.method[2] >= 0 && s->sel == 0) {
        std::string err;
        // (the overlapped form only where PCL_HALO_OVERLAP is set explicitly: with the round's marching y phase and the
        // interior-first order it measures SLOWER than the exchange in front -- 4096 x 2048 Euler block, self-neighbours:
        // no comm 0.675, exchange in front 0.726, overlapped 0.927 ms per step; dense state 1.111 / 1.145 / 1.241)
        const bool ov = s->halo.active && s->overlap && !s->overlap_dflt;    // else: everything on the solver stream, in order
        const bool seq = !ov || s->overlap == 2;
        const hipStream_t hs = seq ? s->stream : s->hstream;
        int rc = PCL_OK;
        if (ov && seq) rc = unsplit_phase(s, 1, dt, 1, s->stream);
        else if (ov) {
            hipError_t he = hipEventRecord(s->ev_h0, s->stream);      // q of the previous step is complete
            if (he == hipSuccess) he = hipStreamWaitEvent(hs, s->ev_h0, 0);
            if (he != hipSuccess) rc = fail(PCL_EHIP, std::string("halo stream fork: ") + hipGetErrorString(he));
        }
        if (!rc && ov && !seq) rc = unsplit_phase(s, 1, dt, 1, s->stream);   // interior tiles first (see the dim-split step)
        if (!rc && s->halo.active && s->halo.exchange(s->q, s->cfg.meqn, s->pitch, s->plane, err, hs))
            rc = fail(PCL_ECOMM, err);
        if (!rc) rc = unsplit_frame(s, hs, bc, cstate);                // all four sides + the copy to t1: one launch
        if (!rc) rc = unsplit_phase(s, 1, dt, ov ? 2 : 0, hs);        // rim tiles behind the frame (or all tiles)
        if (ov && !seq) {                                              // join, also on the error paths
            hipError_t he = hipEventRecord(s->ev_h1, hs);
            if (he == hipSuccess) he = hipStreamWaitEvent(s->stream, s->ev_h1, 0);
            if (!rc && he != hipSuccess) rc = fail(PCL_EHIP, std::string("halo stream join: ") + hipGetErrorString(he));
        }
        if (!rc) rc = unsplit_phase(s, 2, dt, 0, s->stream);
        if (rc) return bail(s, rc);
        std::swap(s->q, s->t1);
        s->undo_slot = &s->t1;
        return read_cfl(s, cfl);
    }
    if (s->halo.active && !overlapped) {
        std::string err;
        if (s->halo.exchange(cur(s), s->cfg.meqn, s->pitch, s->plane, err)) return bail(s, fail(PCL_ECOMM, err));
    }
    if (overlapped) {
        for (int k = 0; k < 4; k++) {
            s->vbc[k] = bc[k];
            for (int m = 0; m < 8; m++)
                s->vconst[k][m] = (s->vbc[k] == PCL_BC_CUSTOM) ? cstate[k * PCL_MAX_RP_PARAMS + m] : 0.0;
        }
        std::string err;
        int rc;
        s->vbc_on = 1;
        // the x pass of the two-pass step, or the whole step where one kernel does both sweeps
        auto first = [&](int sub, hipStream_t on) {
            return onek ? do_step2ds(s, s->q, s->t2, dt, sub, box, on) : do_sweep(s, s->q, s->t1, 1, dt, sub, box, on);
        };
        if (onek) s->ghosts_drop(s->t2);             // the step writes that buffer's interior: its frame is stale
        if (onek && s->overlap == 1 && s->exchange_ahead && s->ghosts_filled(s->q)) {
            // One kernel per step and the halo of THIS state already sent ahead: the rim tiles (everything a neighbour
            // needs of the new state, and everything that reads the ghost frame) go first, on the halo stream behind the
            // exchange that filled the frame; the new state's halo follows them there at once -- it travels while the
            // interior tiles, on the solver stream, are still at work, and the next step's rim tiles queue up behind
            // it.  The step ends when both launches have finished (not the exchange).  A rejected step leaves a stale
            // exchange on the halo stream: it touches only rim cells and the ghost frame of the buffer the retake's
            // interior launch writes elsewhere, and the retake's rim launch is ordered behind it on that stream.
            rc = first(2, s->hstream);
            hipError_t he = hipEventRecord(s->ev_h1, s->hstream);
            if (!rc) rc = first(1, nullptr);
            if (!rc && s->halo.exchange(s->t2, s->cfg.meqn, s->pitch, s->plane, err, s->hstream)) rc = fail(PCL_ECOMM, err);
            if (he == hipSuccess) he = hipStreamWaitEvent(s->stream, s->ev_h1, 0);
            if (!rc && he != hipSuccess) rc = fail(PCL_EHIP, std::string("halo stream join: ") + hipGetErrorString(he));
            s->vbc_on = 0;
            if (rc) { s->ghosts_drop_all(); return bail(s, rc); }
            s->form_steps[1]++;
            std::swap(s->q, s->t2);
            s->undo_slot = &s->t2;
            s->ghosts_mark(s->q);
            return read_cfl(s, cfl);
        }
        if (onek && s->overlap == 1 && !s->exchange_ahead) {
            // One kernel per step, no exchange-ahead: the exchange runs in front of the step on the solver stream and
            // the step is ONE launch.  (An interior launch beside the exchange and a rim launch behind it, the two-pass
            // step's scheme, costs this kernel more than it hides: it fills every CU's LDS and registers, the rim
            // workgroups and the pack / unpack kernels next to it are starved -- 80 and 40-76 us instead of 25 and 6 --
            // and every hand-over between the two hardware queues adds 30-80 us.)
            // Behind the exchange the block steps like a single one: ONE launch or, where every cell is active, the x
            // pass + y pass -- the faster form, re-measured every 256 steps by this rank for itself (pcl_step_hyperbolic;
            // 4096^2 block, dense state: one kernel 1.01 ms, two passes 0.86).
            rc = s->halo.exchange(s->q, s->cfg.meqn, s->pitch, s->plane, err) ? fail(PCL_ECOMM, err) : PCL_OK;
            if (rc) { s->vbc_on = 0; s->ghosts_drop_all(); return bail(s, rc); }
            s->form_seq_tune = 1;
            rc = pcl_step_hyperbolic(s, dt, cfl);       // swaps the buffers, reads the (all-reduced) Courant number
            s->form_seq_tune = 0;
            s->vbc_on = 0;
            return rc;
        }
        if (s->overlap == 2) {
            // test mode (PCL_HALO_OVERLAP=2): same launches on ONE stream with the interior tiles strictly
            // BEFORE the exchange -- an interior tile that read a ghost cell would see the stale frame
            rc = first(1, nullptr);
            if (!rc && s->halo.exchange(s->q, s->cfg.meqn, s->pitch, s->plane, err)) rc = fail(PCL_ECOMM, err);
        } else {
            // exchange-ahead: the previous step (or a retaken step's first attempt) already exchanged this buffer's
            // halo on the halo stream; the rim tiles queue up behind it there
            const bool have = s->exchange_ahead && s->ghosts_filled(s->q);
            hipError_t hf = hipSuccess;
            if (!have) {
                hf = hipEventRecord(s->ev_h0, s->stream);             // q of the previous step is complete
                if (hf == hipSuccess) hf = hipStreamWaitEvent(s->hstream, s->ev_h0, 0);
            }
            if (hf != hipSuccess) {
                s->vbc_on = 0;
                return bail(s, fail(PCL_EHIP, std::string("halo stream fork: ") + hipGetErrorString(hf)));
            }
            // The interior tiles go FIRST: the step starts on an idle device (the host has just read the previous
            // Courant number), and the host needs some tens of microseconds to enqueue the group of Send/Recv -- with
            // the interior launch already queued the device works through that time instead of waiting for it
            rc = first(1, nullptr);      // interior tiles, concurrent with the exchange
            if (!rc && !have) {
                if (s->halo.exchange(s->q, s->cfg.meqn, s->pitch, s->plane, err, s->hstream)) rc = fail(PCL_ECOMM, err);
                else s->ghosts_mark(s->q);
            }
            // rim tiles (ghost frame + physical BCs) behind the exchange on ITS stream: they start as soon as
            // the frame has arrived and fill the machine next to the interior kernel's tail
            if (!rc) rc = first(2, s->hstream);
            hipError_t he = hipEventRecord(s->ev_h1, s->hstream);
            if (he == hipSuccess) he = hipStreamWaitEvent(s->stream, s->ev_h1, 0);
            if (!rc && he != hipSuccess) rc = fail(PCL_EHIP, std::string("halo stream join: ") + hipGetErrorString(he));
        }
        if (!rc && s->overlap == 2) rc = first(2, nullptr);
        s->vbc_on = 0;
        if (rc) { s->ghosts_drop_all(); return bail(s, rc); }
        if (!onek) {
            s->ghosts_drop(s->t2);                   // the y pass writes that buffer, ghost rows included
            if (int rc2 = do_sweep(s, s->t1, s->t2, 2, dt)) { s->ghosts_drop_all(); return bail(s, rc2); }
        }
        s->form_steps[onek ? 1 : 0]++;
        std::swap(s->q, s->t2);
        s->undo_slot = &s->t2;
        if (s->exchange_ahead && s->overlap == 1) {
            // The new state's halo travels NOW, behind the y pass, on the halo stream: through the Courant number's
            // hand-over and the host's decision.  If the step is rejected the exchange was for nothing -- the pre-step
            // buffer comes back (pcl_undo_step) with its own ghost frame still filled.  The CFL all-reduce is
            // enqueued FIRST (read_cfl's issue order on the communicator: all-reduce, then this group, on every rank).
            const int rc3 = read_cfl_begin(s);
            if (rc3) { s->ghosts_drop_all(); return rc3; }
            hipError_t he = hipEventRecord(s->ev_y, s->stream);
            if (he == hipSuccess) he = hipStreamWaitEvent(s->hstream, s->ev_y, 0);
            if (he != hipSuccess) { s->ghosts_drop_all(); return bail(s, fail(PCL_EHIP, std::string("exchange-ahead fork: ") + hipGetErrorString(he))); }
            if (s->halo.exchange(s->q, s->cfg.meqn, s->pitch, s->plane, err, s->hstream)) {
                s->ghosts_drop_all();
                return bail(s, fail(PCL_ECOMM, err));
            }
            s->ghosts_mark(s->q);
            return read_cfl_end(s, cfl);
        }
        return read_cfl(s, cfl);
    }
    if (fused) {
        for (int k = 0; k < 4; k++) {
            s->vbc[k] = k < 2 * s->cfg.ndim ? bc[k] : -1;
            for (int m = 0; m < 8; m++)
                s->vconst[k][m] = (s->vbc[k] == PCL_BC_CUSTOM) ? cstate[k * PCL_MAX_RP_PARAMS + m] : 0.0;
        }
        s->vbc_on = 1;
        s->form_seq_tune = s->halo.active ? 1 : 0;      // a decomposed block behind its exchange: the faster form, too
        const int rc = pcl_step_hyperbolic(s, dt, cfl);
        s->form_seq_tune = 0;
        s->vbc_on = 0;
        return rc;
    }
    for (int idim = 0; idim < s->cfg.ndim; idim++)
        for (int side = 0; side < 2; side++) {
            const int t = bc[2 * idim + side];
            if (t < 0) continue;
            const int rc = t == PCL_BC_CUSTOM
                               ? bc_launch(s, idim, side, 100, cstate + (2 * idim + side) * PCL_MAX_RP_PARAMS)
                               : bc_launch(s, idim, side, t, nullptr);
            if (rc) return rc;
        }
    return pcl_step_hyperbolic(s, dt, cfl);
}

int pcl_undo_step(pcl_solver *s) {
    if (!s) return fail(PCL_EINVAL, "null argument");
    if (!s->undo_slot) return fail(PCL_ESTATE, "no step to undo");
    std::swap(s->q, *s->undo_slot);
    s->undo_slot = nullptr;
    return PCL_OK;
}

int pcl_backup(pcl_solver *s) {
    if (!s) return fail(PCL_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    const size_t qbytes = (size_t)s->total * sizeof(double);
    if (!s->bak) {
        double *raw = nullptr;
        HIP_TRY(hipMalloc((void **)&raw, qbytes + 16 * sizeof(double)));
        s->bak = raw + s->lead;
    }
    HIP_TRY(hipMemcpyAsync(s->bak, s->q, qbytes, hipMemcpyDeviceToDevice, s->stream));
    return PCL_OK;
}

int pcl_restore(pcl_solver *s) {
    if (s) s->ghosts_drop_all();          // exchange-ahead: whatever filled the ghost frames no longer holds
    if (!s) return fail(PCL_EINVAL, "null argument");
    if (!s->bak) return fail(PCL_ESTATE, "pcl_restore without pcl_backup");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipMemcpyAsync(s->q, s->bak, (size_t)s->total * sizeof(double), hipMemcpyDeviceToDevice,
                           s->stream));
    s->undo_slot = nullptr;
    return PCL_OK;
}

int pcl_src(pcl_solver *s, int src_id, double dt, const double *params, int nparams) {
    if (s) s->ghosts_drop_all();          // exchange-ahead: whatever filled the ghost frames no longer holds
    if (!s) return fail(PCL_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    if (src_id == PCL_SRC_EULER_RADIAL) {
        if (s->cfg.rp != PCL_RP_EULER5_2D || s->cfg.maux < 1 || nparams < 2 || !params)
            return fail(PCL_EINVAL, "euler radial source needs the Euler solver, aux[0]=radius, {gamma1,ndim}");
        dim3 grid((s->cfg.n[0] + 255) / 256, s->cfg.n[1]);
        hipLaunchKernelGGL(src_euler_radial, grid, dim3(256), 0, s->stream, s->q, s->aux, s->cfg.mbc,
                           s->cfg.n[0], s->cfg.n[1], s->pitch, s->plane, dt, params[0], params[1] - 1.0);
        HIP_TRY(hipGetLastError());
        return PCL_OK;
    }
    if (src_id == PCL_SRC_SPHERE_CORIOLIS) {
        if (s->cfg.meqn != 4 || s->cfg.maux < 16 || s->cfg.ndim != 2 || !s->aux)
            return fail(PCL_EINVAL, "sphere Coriolis source needs q = (h,hu,hv,hw) and the 16 aux components of setaux.f");
        dim3 grid((s->cfg.n[0] + 255) / 256, s->cfg.n[1]);
        hipLaunchKernelGGL(src_sphere_coriolis, grid, dim3(256), 0, s->stream, cur(s), s->aux, s->cfg.mbc, s->cfg.n[0],
                           s->cfg.n[1], s->pitch, s->plane, dt);
        HIP_TRY(hipGetLastError());
        return PCL_OK;
    }
    return fail(PCL_EINVAL, "unknown source id");
}

int pcl_fuse_source(pcl_solver *s, int src_id, const double *params, int nparams) {
    if (!s) return fail(PCL_EINVAL, "null argument");
    if (src_id == 0) { s->fused_src = 0; return PCL_OK; }
    if (src_id != PCL_SRC_EULER_RADIAL) return fail(PCL_EINVAL, "pcl_fuse_source: only the Euler radial source can be fused");
    if (s->cfg.kind != PCL_KIND_CLASSIC || s->cfg.ndim != 2 || s->cfg.rp != PCL_RP_EULER5_2D || s->cfg.maux < 1 ||
        s->cfg.method[5] != 0)
        return fail(PCL_EINVAL, "pcl_fuse_source: 2-D Euler solver without a capacity function, radial coordinate in aux(1)");
    if (!params || nparams < 2) return fail(PCL_EINVAL, "Euler radial source needs (gamma1, ndim)");
    s->fused_src = src_id;
    s->fused_src_p[0] = params[0];
    s->fused_src_p[1] = params[1] - 1.0;
    return PCL_OK;
}

int pcl_sharp_fuse_dq_src(pcl_solver *s, int src_id, const double *params, int nparams) {
    if (!s) return fail(PCL_EINVAL, "null argument");
    if (s->cfg.kind != PCL_KIND_SHARPCLAW) return fail(PCL_EINVAL, "pcl_sharp_fuse_dq_src: not a SharpClaw solver");
    if (src_id == 0) { s->fused_src = 0; return PCL_OK; }
    if (src_id != PCL_SRC_EULER_RADIAL) return fail(PCL_EINVAL, "pcl_sharp_fuse_dq_src: only the Euler radial source can be fused");
    if (s->cfg.ndim != 2 || s->cfg.rp != PCL_RP_EULER5_2D || s->cfg.maux < 1 || s->cfg.method[5] != 0 ||
        s->cfg.mbc != 3 || s->cfg.lim_type != 2)
        return fail(PCL_EINVAL, "pcl_sharp_fuse_dq_src: 2-D Euler solver, WENO5 (lim_type 2, mbc 3), no capacity function, "
                                "radial coordinate in aux(1)");
    if (!params || nparams < 2) return fail(PCL_EINVAL, "Euler radial source needs (gamma1, ndim)");
    s->fused_src = src_id;
    s->fused_src_p[0] = params[0];
    s->fused_src_p[1] = params[1] - 1.0;
    return PCL_OK;
}

int pcl_select(pcl_solver *s, int reg) {
    if (s) s->ghosts_drop_all();          // exchange-ahead: whatever filled the ghost frames no longer holds
    if (!s) return fail(PCL_EINVAL, "null argument");
    if (reg == 0) { s->sel = 0; return PCL_OK; }
    if (s->cfg.kind != PCL_KIND_SHARPCLAW || reg < 0 || reg > 4) return fail(PCL_EINVAL, "bad register");
    s->sel = reg;
    return PCL_OK;
}

// one directional pass of a stage; sub: 0 all tiles, 1 tiles that read no ghost cell, 2 the others (x pass only)
static int sharp_pass(pcl_solver *s, int ids, double dt, int rk_op, const double *ra, const double *rb, double *rd,
                      double ca, double cb, double cc, int sub, hipStream_t stream) {
    std::string err;
    SweepLaunch l;
    l.a = make_args(s, cur(s), s->sreg[PCL_REG_DQ], ids, dt);
    l.a.sub = sub;
    if (rk_op && ids == s->cfg.ndim) {
        l.a.rk_op = rk_op; l.a.rk_a = ra; l.a.rk_b = rb; l.a.rk_d = rd;
        l.a.rk_ca = ca; l.a.rk_cb = cb; l.a.rk_cc = cc;
    }
    l.ndim = s->cfg.ndim; l.rp = s->cfg.rp; l.ids = ids; l.fwave = s->cfg.fwave;
    l.lim_type = s->cfg.lim_type; l.stream = stream;
    l.char_decomp = s->cfg.method[4];          // SharpClaw: clawparams.char_decomp travels in method(5) (unused by it)
    pcl_solver::Timed t{};
    if (timing_on(s)) { t.a = get_event(s); t.b = get_event(s); t.which = ids - 1; t.count = sub != 2; hipEventRecord(t.a, stream); }
    int rc = PCL_BY_MATH(s->cfg.math, launch_sharp(l, err));
    if (timing_on(s)) { hipEventRecord(t.b, stream); s->timed.push_back(t); if (s->timed.size() >= 2048) drain_timing(s); }
    if (rc) return fail(rc, err);
    return PCL_OK;
}

// ghost frame of the selected register: halo exchange (decomposed runs), then the physical BCs in the reference's
// order (per dimension, lower then upper); bc[k] < 0: nothing to do on that side
static int sharp_frame(pcl_solver *s, const int *bc, const double *cstate, hipStream_t stream) {
    std::string err;
    if (s->halo.active && s->halo.exchange(cur(s), s->cfg.meqn, s->pitch, s->plane, err, stream))
        return fail(PCL_ECOMM, err);
    if (s->cfg.ndim == 2) {       // all four sides in one launch (the index-remap composition of frame_kernel)
        FrameBc f;
        f.only = 0;
        for (int k = 0; k < 4; k++) {
            f.t[k] = bc[k] == PCL_BC_CUSTOM ? 100 : bc[k];
            for (int m = 0; m < 8; m++) f.c[k][m] = (bc[k] == PCL_BC_CUSTOM && cstate) ? cstate[k * PCL_MAX_RP_PARAMS + m] : 0.0;
        }
        hipLaunchKernelGGL(frame_kernel, dim3(frame_blocks(s)), dim3(256), 0, stream, cur(s), (double *)nullptr, s->cfg.meqn,
                           s->I, s->J, s->cfg.mbc, s->pitch, s->plane, f);
        HIP_TRY(hipGetLastError());
        return PCL_OK;
    }
    for (int idim = 0; idim < s->cfg.ndim; idim++)
        for (int side = 0; side < 2; side++) {
            const int t = bc[2 * idim + side];
            if (t < 0) continue;
            if (t == PCL_BC_SPHERE_MIRROR) {
                const long n = (long)s->I * s->cfg.mbc * s->cfg.meqn;
                hipLaunchKernelGGL(bc_sphere_mirror, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, cur(s),
                                   s->cfg.meqn, s->I, s->J, s->pitch, s->plane, s->cfg.mbc, side);
                HIP_TRY(hipGetLastError());
                continue;
            }
            const int rc = t == PCL_BC_CUSTOM
                               ? bc_launch(s, idim, side, 100, cstate + (2 * idim + side) * PCL_MAX_RP_PARAMS, false, stream)
                               : bc_launch(s, idim, side, t, nullptr, false, stream);
            if (rc) return rc;
        }
    return PCL_OK;
}

// The passes of one stage.  bc != nullptr: the stage's ghost frame is built here too; in a decomposed 2-D run it is
// built on the halo stream while the x pass runs the tiles that read no ghost cell (the rim tiles follow on the halo
// stream, the y pass after the join) -- the SharpClaw counterpart of pcl_bc_step's choreography, ten times per
// SSP104 step.  PCL_HALO_OVERLAP=2: same launches on one stream, interior tiles strictly BEFORE the frame.
static int sharp_passes(pcl_solver *s, double dt, int rk_op, const double *ra, const double *rb, double *rd,
                        double ca, double cb, double cc, const int *bc = nullptr, const double *cstate = nullptr) {
    // the fused store phase dereferences all three operands whenever rk_op != 0 (sharpclaw.hpp): a null one is a
    // memory fault at address 0 on the device, so it is refused here
    if (rk_op && (!ra || !rb || !rd)) return fail(PCL_EINVAL, "fused RK stage: null register operand");
    if (!cur(s) || !s->sreg[PCL_REG_DQ]) return fail(PCL_ESTATE, "SharpClaw registers not allocated");
    int rc = PCL_OK;
    int first = 1;
    if (bc && s->halo.active && s->overlap && s->cfg.ndim == 2) {
        const bool seq = s->overlap == 2;
        const hipStream_t hs = seq ? s->stream : s->hstream;
        if (seq) rc = sharp_pass(s, 1, dt, rk_op, ra, rb, rd, ca, cb, cc, 1, s->stream);
        else {
            hipError_t he = hipEventRecord(s->ev_h0, s->stream);      // the stage register is complete
            if (he == hipSuccess) he = hipStreamWaitEvent(hs, s->ev_h0, 0);
            if (he != hipSuccess) rc = fail(PCL_EHIP, std::string("halo stream fork: ") + hipGetErrorString(he));
        }
        if (!rc && !seq) rc = sharp_pass(s, 1, dt, rk_op, ra, rb, rd, ca, cb, cc, 1, s->stream);   // interior tiles first
        if (!rc) rc = sharp_frame(s, bc, cstate, hs);
        if (!rc) rc = sharp_pass(s, 1, dt, rk_op, ra, rb, rd, ca, cb, cc, 2, hs);
        if (!seq) {                                                    // join, also on the error paths
            hipError_t he = hipEventRecord(s->ev_h1, hs);
            if (he == hipSuccess) he = hipStreamWaitEvent(s->stream, s->ev_h1, 0);
            if (!rc && he != hipSuccess) rc = fail(PCL_EHIP, std::string("halo stream join: ") + hipGetErrorString(he));
        }
        first = 2;
    } else if (bc) {
        rc = sharp_frame(s, bc, cstate, s->stream);
    }
    for (int ids = first; ids <= s->cfg.ndim && !rc; ids++)
        rc = sharp_pass(s, ids, dt, rk_op, ra, rb, rd, ca, cb, cc, 0, s->stream);
    return rc ? bail(s, rc) : PCL_OK;
}

static int check_bc_spec(pcl_solver *s, const int *bc, const double *cstate) {
    for (int k = 0; k < 2 * s->cfg.ndim; k++) {
        const int t = bc[k];
        if (t == PCL_BC_SPHERE_MIRROR) {
            if (s->cfg.ndim != 2 || k < 2) return fail(PCL_EINVAL, "PCL_BC_SPHERE_MIRROR is the y boundary of a 2-D grid");
            continue;
        }
        if (t >= 0 && t != PCL_BC_CUSTOM && t != PCL_BC_OUTFLOW && t != PCL_BC_PERIODIC && t != PCL_BC_REFLECTING)
            return fail(PCL_EINVAL, "bad boundary condition type");
        if (t == PCL_BC_CUSTOM && !cstate) return fail(PCL_EINVAL, "constant state missing");
    }
    return PCL_OK;
}

static int sharp_dq_impl(pcl_solver *s, const int *bc, const double *cstate, double dt, double *cfl) {
    if (!s || !cfl) return fail(PCL_EINVAL, "null argument");
    if (s->cfg.kind != PCL_KIND_SHARPCLAW) return fail(PCL_ESTATE, "SharpClaw call on a classic solver");
    if (s->sel == PCL_REG_DQ) return fail(PCL_EINVAL, "dq of the dq register");
    if (bc) if (int rc = check_bc_spec(s, bc, cstate)) return rc;
    HIP_TRY(hipSetDevice(s->cfg.device));
    if (int rc = sharp_passes(s, dt, 0, nullptr, nullptr, nullptr, 0, 0, 0, bc, cstate)) return rc;
    return read_cfl(s, cfl);
}

int pcl_sharp_dq(pcl_solver *s, double dt, double *cfl) { return sharp_dq_impl(s, nullptr, nullptr, dt, cfl); }

int pcl_sharp_bc_dq(pcl_solver *s, const int *bc, const double *cstate, double dt, double *cfl) {
    if (!bc) return fail(PCL_EINVAL, "null argument");
    return sharp_dq_impl(s, bc, cstate, dt, cfl);
}

static int sharp_stage_impl(pcl_solver *s, const int *bc, const double *cstate, double dt, int op, int D, int A, int B,
                            double ca, double cb, double cc, double cfl_max, double *cfl) {
    if (!s || !cfl) return fail(PCL_EINVAL, "null argument");
    if (s->cfg.kind != PCL_KIND_SHARPCLAW) return fail(PCL_ESTATE, "SharpClaw call on a classic solver");
    if (op != 1 && op != 2 && op != 5) return fail(PCL_EINVAL, "pcl_sharp_stage fuses RK ops 1, 2 and 5");
    if (s->sel == PCL_REG_DQ || s->sel == PCL_REG_TMP) return fail(PCL_EINVAL, "stage of the dq/tmp register");
    auto reg = [&](int r) -> double *& { return r == 0 ? s->q : s->sreg[r]; };
    if (D < 0 || D > 2 || A < 0 || A > 2 || B < 0 || B > 2) return fail(PCL_EINVAL, "registers must be q, s1 or s2");
    HIP_TRY(hipSetDevice(s->cfg.device));
    // the result goes to the spare register and becomes D by a pointer swap once the stage's Courant number is
    // known to be acceptable: D may be the register the passes read with their halo, or the state itself
    double *spare = s->sreg[PCL_REG_TMP];
    if (!spare || !reg(A) || !reg(B) || !reg(D) || !s->sreg[PCL_REG_DQ])
        return fail(PCL_ESTATE, "pcl_sharp_stage: register not allocated (A=" + std::to_string(A) + " B=" + std::to_string(B) +
                                    " D=" + std::to_string(D) + ")");
    // op 1 does not use B; the kernel still loads it (its store phase is branch-free, sharpclaw.hpp): point it at A
    // so that the load hits the line A's load just brought in instead of streaming another array
    if (bc) if (int rc = check_bc_spec(s, bc, cstate)) return rc;
    if (int rc = sharp_passes(s, dt, op, reg(A), op == 1 ? reg(A) : reg(B), spare, ca, cb, cc, bc, cstate)) return rc;
    if (int rc = read_cfl(s, cfl)) return rc;
    if (*cfl <= cfl_max) {
        std::swap(reg(D), s->sreg[PCL_REG_TMP]);
        s->undo_slot = nullptr;
    }
    return PCL_OK;
}

int pcl_sharp_stage(pcl_solver *s, double dt, int op, int D, int A, int B, double ca, double cb, double cc,
                    double cfl_max, double *cfl) {
    return sharp_stage_impl(s, nullptr, nullptr, dt, op, D, A, B, ca, cb, cc, cfl_max, cfl);
}

int pcl_sharp_bc_stage(pcl_solver *s, const int *bc, const double *cstate, double dt, int op, int D, int A, int B,
                       double ca, double cb, double cc, double cfl_max, double *cfl) {
    if (!bc) return fail(PCL_EINVAL, "null argument");
    return sharp_stage_impl(s, bc, cstate, dt, op, D, A, B, ca, cb, cc, cfl_max, cfl);
}

int pcl_rk_op(pcl_solver *s, int op, int D, int A, int B, int Cc, double ca, double cb, double cc) {
    if (!s) return fail(PCL_EINVAL, "null argument");
    if (s->cfg.kind != PCL_KIND_SHARPCLAW) return fail(PCL_ESTATE, "SharpClaw call on a classic solver");
    if (op < 1 || op > 6) return fail(PCL_EINVAL, "unknown RK op");
    if (op == 6 && (Cc == A || Cc == B || D != B)) return fail(PCL_EINVAL, "op 6 writes C and overwrites B: C must differ from A and B, D must be B");
    auto reg = [&](int r) -> double * { return r == 0 ? s->q : ((r > 0 && r < 5) ? s->sreg[r] : nullptr); };
    RkLaunch r;
    r.d = reg(D); r.a = reg(A); r.b = reg(B); r.c = reg(Cc);
    if (!r.d || !r.a || !r.b || !r.c) return fail(PCL_EINVAL, "bad register");
    r.ca = ca; r.cb = cb; r.cc = cc; r.op = op;
    r.n = s->total;
    HIP_TRY(hipSetDevice(s->cfg.device));
    std::string err;
    int rc = PCL_BY_MATH(s->cfg.math, launch_rk(r, s->stream, err));
    if (rc) return fail(rc, err);
    return PCL_OK;
}

int pcl_sync(pcl_solver *s) {
    if (!s) return fail(PCL_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return PCL_OK;
}

int pcl_timer_start(pcl_solver *s) {
    if (!s) return fail(PCL_EINVAL, "null argument");
    HIP_TRY(hipEventRecord(s->ev0, s->stream));
    return PCL_OK;
}
int pcl_timer_stop(pcl_solver *s, float *ms) {
    if (!s || !ms) return fail(PCL_EINVAL, "null argument");
    HIP_TRY(hipEventRecord(s->ev1, s->stream));
    HIP_TRY(hipEventSynchronize(s->ev1));
    HIP_TRY(hipEventElapsedTime(ms, s->ev0, s->ev1));
    return PCL_OK;
}
int pcl_kernel_timing(pcl_solver *s, int enable) {
    if (!s) return fail(PCL_EINVAL, "null argument");
    if (int rc = drain_timing(s)) return rc;
    s->timing = enable < 0 ? 0 : enable;
    s->kt_ms[0] = s->kt_ms[1] = s->kt_ms[2] = 0;
    s->kt_n[0] = s->kt_n[1] = s->kt_n[2] = 0;
    s->form_steps[0] = s->form_steps[1] = 0;
    return PCL_OK;
}
int pcl_step_count(pcl_solver *s, long *steps) {
    if (!s || !steps) return fail(PCL_EINVAL, "null argument");
    *steps = s->step_no;
    return PCL_OK;
}
int pcl_kernel_timing_read(pcl_solver *s, double *ms_total, long *launches) {
    if (!s || !ms_total || !launches) return fail(PCL_EINVAL, "null argument");
    if (int rc = drain_timing(s)) return rc;
    ms_total[0] = s->kt_ms[0]; ms_total[1] = s->kt_ms[1];
    launches[0] = s->kt_n[0]; launches[1] = s->kt_n[1];
    return PCL_OK;
}

int pcl_step_form_stats(pcl_solver *s, double *ms_total, long *launches, long *steps_one_kernel, long *steps_two_pass) {
    if (!s || !ms_total || !launches || !steps_one_kernel || !steps_two_pass) return fail(PCL_EINVAL, "null argument");
    if (int rc = drain_timing(s)) return rc;
    *ms_total = s->kt_ms[2];
    *launches = s->kt_n[2];
    *steps_one_kernel = s->form_steps[1];
    *steps_two_pass = s->form_steps[0];
    return PCL_OK;
}

int pcl_debug_wave_shift(const double *in64, double *left64, double *right64) {
    if (int rc = check_device()) return rc;
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 3 * 64 * sizeof(double)));
    HIP_TRY(hipMemcpy(d, in64, 64 * sizeof(double), hipMemcpyHostToDevice));
    pcl::exact::launch_shift_test(d, d + 64, d + 128);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(left64, d + 64, 64 * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(right64, d + 128, 64 * sizeof(double), hipMemcpyDeviceToHost));
    hipFree(d);
    return PCL_OK;
}

// ---- layer 1: f2py-shaped host calls -------------------------------------------------------------
// The reference's f2py modules are stateless; so are these entry points for the caller.  Inside, the last solver
// handle is kept and reused while the array shapes stay the same (a time loop calls with identical shapes every
// step): no hipMalloc / hipFree per call, only the PCIe transfers the host-array interface implies.
// pcl_layer1_release() frees it.
namespace {
// No destructor: at process exit the HIP runtime may already be torn down when static destructors of this library run
// (it registers its own teardown lazily, i.e. AFTER this library was loaded), so the handle is released explicitly --
// pcl_layer1_release(), which pyclaw_amd._lib registers with atexit -- or left to the process teardown.
struct Layer1Cache {
    pcl_solver *s = nullptr;
};
Layer1Cache g_l1;
// the f2py-shaped calls share that one handle: one caller at a time (the reference's f2py modules hold the GIL and
// module-level work arrays the same way)
std::mutex g_l1_mutex;

bool same_shape(const pcl_config &a, const pcl_config &b) {
    return a.ndim == b.ndim && a.n[0] == b.n[0] && a.n[1] == b.n[1] && a.n[2] == b.n[2] && a.mbc == b.mbc &&
           a.meqn == b.meqn && a.mwaves == b.mwaves && a.maux == b.maux && a.rp == b.rp && a.kind == b.kind &&
           a.device == b.device && a.math == b.math && a.lim_type == b.lim_type && a.fwave == b.fwave &&
           a.method[5] == b.method[5] && (a.method[2] < 0) == (b.method[2] < 0) && a.method[4] == b.method[4];
}
// a handle for this configuration: the cached one (its scalars refreshed) or a new one that replaces it
int g_l1_math = PCL_MATH_EXACT;     // arithmetic mode of the f2py-shaped calls (pcl_layer1_math)
int layer1_handle(const pcl_config &c, pcl_solver **out) {
    if (g_l1.s && same_shape(g_l1.s->cfg, c)) {
        g_l1.s->cfg = c;      // method, mthlim, rp_params, d: read at launch time only
        g_l1.s->undo_slot = nullptr;
        g_l1.s->sel = 0;
        *out = g_l1.s;
        return PCL_OK;
    }
    if (g_l1.s) { pcl_destroy(g_l1.s); g_l1.s = nullptr; }
    if (int rc = pcl_create(&c, &g_l1.s)) { g_l1.s = nullptr; return rc; }
    *out = g_l1.s;
    return PCL_OK;
}
}  // namespace

int pcl_layer1_math(int math) {
    if (math != PCL_MATH_EXACT && math != PCL_MATH_FAST && math != PCL_MATH_STRICT) return fail(PCL_EINVAL, "unknown math mode");
    std::lock_guard<std::mutex> lock(g_l1_mutex);
    g_l1_math = math;
    return PCL_OK;
}

static void layer1_release_locked() {
    if (g_l1.s) { pcl_destroy(g_l1.s); g_l1.s = nullptr; }
}

void pcl_layer1_release(void) {
    std::lock_guard<std::mutex> lock(g_l1_mutex);
    layer1_release_locked();
}

static int host_sweep(int ndim, int rp, const double *rp_params, int fwave, int meqn, int mwaves,
                      int maux, int mbc, int mx, int my, const double *qold, double *qnew,
                      const double *aux, double dx, double dy, double dt, const int *method,
                      const int *mthlim, double *cfl, int ids, bool unsplit) {
    if (!qold || !qnew || !method || !mthlim || !cfl) return fail(PCL_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(g_l1_mutex);
    pcl_config c;
    memset(&c, 0, sizeof(c));
    c.ndim = ndim; c.n[0] = mx; c.n[1] = my; c.mbc = mbc; c.meqn = meqn; c.mwaves = mwaves;
    c.maux = method[6];
    (void)maux;
    for (int k = 0; k < 7; k++) c.method[k] = method[k];
    if (mwaves < 1 || mwaves > PCL_MAX_WAVES) return fail(PCL_EINVAL, "bad mwaves");
    for (int k = 0; k < mwaves; k++) c.mthlim[k] = mthlim[k];
    c.fwave = fwave; c.rp = rp;
    if (rp_params) for (int k = 0; k < PCL_MAX_RP_PARAMS; k++) c.rp_params[k] = rp_params[k];
    c.d[0] = dx; c.d[1] = dy; c.device = 0; c.math = g_l1_math;
    // The Fortran updates qnew IN PLACE (qnew = qnew + increments computed from qold).  Both callers of the reference
    // pass qnew == qold on entry -- a copy (clawpack.py:529-530,538-541) or the same array (:542-543); the device path
    // starts from qold, so anything else is refused instead of silently ignored.
    if (qnew != qold) {
        const size_t nq = (size_t)meqn * (mx + 2 * mbc) * (ndim > 1 ? my + 2 * mbc : 1);
        if (memcmp(qold, qnew, nq * sizeof(double)) != 0)
            return fail(PCL_EINVAL, "qnew must equal qold on entry (a copy or the same array), as in the reference's callers");
    }
    pcl_solver *s = nullptr;
    if (int rc = layer1_handle(c, &s)) return rc;
    int rc = pcl_put_q(s, qold, 1);
    if (!rc && c.maux > 0) rc = aux ? pcl_put_aux(s, aux) : fail(PCL_EINVAL, "aux missing");
    if (!rc) rc = unsplit ? pcl_step_hyperbolic(s, dt, cfl) : pcl_sweep(s, ids, dt, cfl);
    if (!rc) rc = pcl_get_q(s, qnew, 1);
    if (rc) layer1_release_locked();      // never keep a handle that failed
    return rc;
}

int pcl_step1(int rp, const double *rp_params, int meqn, int mwaves, int maux, int mbc, int mx,
              double *q, const double *aux, double dx, double dt, const int *method,
              const int *mthlim, double *cfl) {
    return host_sweep(1, rp, rp_params, 0, meqn, mwaves, maux, mbc, mx, 1, q, q, aux, dx, 1.0, dt,
                      method, mthlim, cfl, 1, false);
}

int pcl_step1fw(int rp, const double *rp_params, int meqn, int mwaves, int maux, int mbc, int mx,
                double *q, const double *aux, double dx, double dt, const int *method,
                const int *mthlim, double *cfl) {
    return host_sweep(1, rp, rp_params, 1, meqn, mwaves, maux, mbc, mx, 1, q, q, aux, dx, 1.0, dt,
                      method, mthlim, cfl, 1, false);
}

int pcl_step2ds(int rp, const double *rp_params, int fwave, int meqn, int mwaves, int maux, int mbc,
                int mx, int my, const double *qold, double *qnew, const double *aux, double dx,
                double dy, double dt, const int *method, const int *mthlim, double *cfl, int ids) {
    if (ids != 1 && ids != 2) return fail(PCL_EINVAL, "ids must be 1 or 2");
    return host_sweep(2, rp, rp_params, fwave, meqn, mwaves, maux, mbc, mx, my, qold, qnew, aux, dx, dy,
                      dt, method, mthlim, cfl, ids, false);
}

int pcl_step2(int rp, const double *rp_params, int fwave, int meqn, int mwaves, int maux, int mbc,
              int mx, int my, const double *qold, double *qnew, const double *aux, double dx, double dy,
              double dt, const int *method, const int *mthlim, double *cfl) {
    if (!method || method[2] < 0) return fail(PCL_EINVAL, "step2 needs method[2] >= 0 (unsplit)");
    return host_sweep(2, rp, rp_params, fwave, meqn, mwaves, maux, mbc, mx, my, qold, qnew, aux, dx, dy,
                      dt, method, mthlim, cfl, 0, true);
}

int pcl_step3ds(int rp, const double *rp_params, int meqn, int mwaves, int maux, int mbc, int mx, int my, int mz,
                const double *qold, double *qnew, const double *aux, double dx, double dy, double dz, double dt,
                const int *method, const int *mthlim, double *cfl, int idir) {
    if (!qold || !qnew || !method || !mthlim || !cfl) return fail(PCL_EINVAL, "null argument");
    if (idir < 1 || idir > 3) return fail(PCL_EINVAL, "idir must be 1, 2 or 3");
    if (mwaves < 1 || mwaves > PCL_MAX_WAVES) return fail(PCL_EINVAL, "bad mwaves");
    std::lock_guard<std::mutex> lock(g_l1_mutex);
    pcl_config c;
    memset(&c, 0, sizeof(c));
    c.ndim = 3; c.n[0] = mx; c.n[1] = my; c.n[2] = mz; c.mbc = mbc; c.meqn = meqn; c.mwaves = mwaves;
    c.maux = maux;
    for (int k = 0; k < 7; k++) c.method[k] = method[k];
    for (int k = 0; k < mwaves; k++) c.mthlim[k] = mthlim[k];
    c.rp = rp;
    if (rp_params) for (int k = 0; k < PCL_MAX_RP_PARAMS; k++) c.rp_params[k] = rp_params[k];
    c.d[0] = dx; c.d[1] = dy; c.d[2] = dz; c.math = g_l1_math;
    pcl_solver *s = nullptr;
    if (int rc = layer1_handle(c, &s)) return rc;
    int rc = pcl_put_q(s, qold, 1);
    if (!rc && maux > 0) rc = aux ? pcl_put_aux(s, aux) : fail(PCL_EINVAL, "aux missing");
    if (!rc) rc = pcl_sweep(s, idir, dt, cfl);
    if (!rc) rc = pcl_get_q(s, qnew, 1);
    if (rc) layer1_release_locked();
    return rc;
}

// clawparams.mthlim of the F90 module state (sharpclaw.py:268): only lim_type = 1 (tvd2) reads it
static int g_sharp_mthlim[PCL_MAX_WAVES] = {1, 1, 1, 1, 1, 1, 1, 1};

// clawparams.char_decomp of the F90 module state (sharpclaw.py:262): read by pcl_sharp_flux1
static int g_sharp_char_decomp = 0;
int pcl_sharp_module_char_decomp(int char_decomp) {
    if (char_decomp != 0 && char_decomp != 1) return fail(PCL_EINVAL, "pcl_sharp_module_char_decomp: 0 or 1");
    std::lock_guard<std::mutex> lock(g_l1_mutex);
    g_sharp_char_decomp = char_decomp;
    return PCL_OK;
}

int pcl_sharp_module_mthlim(const int *mthlim, int n) {
    if (!mthlim || n < 0 || n > PCL_MAX_WAVES) return fail(PCL_EINVAL, "pcl_sharp_module_mthlim: 0 <= n <= PCL_MAX_WAVES");
    std::lock_guard<std::mutex> lock(g_l1_mutex);
    for (int k = 0; k < PCL_MAX_WAVES; k++) g_sharp_mthlim[k] = k < n ? mthlim[k] : 1;
    return PCL_OK;
}

int pcl_step3(int rp, const double *rp_params, int meqn, int mwaves, int maux, int mbc, int mx, int my, int mz,
              const double *qold, double *qnew, const double *aux, double dx, double dy, double dz, double dt,
              const int *method, const int *mthlim, double *cfl) {
    if (!qold || !qnew || !method || !mthlim || !cfl) return fail(PCL_EINVAL, "null argument");
    if (method[2] < 0) return fail(PCL_EINVAL, "step3 needs method[2] >= 0 (unsplit)");
    if (mwaves < 1 || mwaves > PCL_MAX_WAVES) return fail(PCL_EINVAL, "bad mwaves");
    std::lock_guard<std::mutex> lock(g_l1_mutex);
    pcl_config c;
    memset(&c, 0, sizeof(c));
    c.ndim = 3; c.n[0] = mx; c.n[1] = my; c.n[2] = mz; c.mbc = mbc; c.meqn = meqn; c.mwaves = mwaves;
    c.maux = maux;
    for (int k = 0; k < 7; k++) c.method[k] = method[k];
    for (int k = 0; k < mwaves; k++) c.mthlim[k] = mthlim[k];
    c.rp = rp;
    if (rp_params) for (int k = 0; k < PCL_MAX_RP_PARAMS; k++) c.rp_params[k] = rp_params[k];
    c.d[0] = dx; c.d[1] = dy; c.d[2] = dz; c.math = g_l1_math;
    pcl_solver *s = nullptr;
    if (int rc = layer1_handle(c, &s)) return rc;
    int rc = pcl_put_q(s, qold, 1);
    if (!rc && maux > 0) rc = aux ? pcl_put_aux(s, aux) : fail(PCL_EINVAL, "aux missing");
    if (!rc) rc = pcl_step_hyperbolic(s, dt, cfl);
    if (!rc) rc = pcl_get_q(s, qnew, 1);
    if (rc) layer1_release_locked();
    return rc;
}

static int host_sharp(int ndim, int rp, const double *rp_params, int lim_type, int meqn, int mwaves, int maux,
                      int mcapa, int mbc, int mx, int my, const double *q, double *dq, const double *aux,
                      double dx, double dy, double dt, double *cfl) {
    if (!q || !dq || !cfl) return fail(PCL_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(g_l1_mutex);
    pcl_config c;
    memset(&c, 0, sizeof(c));
    c.ndim = ndim; c.n[0] = mx; c.n[1] = my; c.mbc = mbc; c.meqn = meqn; c.mwaves = mwaves; c.maux = maux;
    c.method[1] = 2; c.method[5] = mcapa; c.method[6] = maux;
    c.method[4] = ndim == 1 ? g_sharp_char_decomp : 0;
    for (int k = 0; k < PCL_MAX_WAVES; k++) c.mthlim[k] = g_sharp_mthlim[k];
    c.rp = rp;
    if (rp_params) for (int k = 0; k < PCL_MAX_RP_PARAMS; k++) c.rp_params[k] = rp_params[k];
    c.d[0] = dx; c.d[1] = dy; c.kind = PCL_KIND_SHARPCLAW; c.lim_type = lim_type; c.math = g_l1_math;
    pcl_solver *s = nullptr;
    if (int rc = layer1_handle(c, &s)) return rc;
    int rc = pcl_put_q(s, q, 1);
    if (!rc && maux > 0) rc = aux ? pcl_put_aux(s, aux) : fail(PCL_EINVAL, "aux missing");
    if (!rc) rc = pcl_sharp_dq(s, dt, cfl);
    if (!rc) rc = pcl_select(s, PCL_REG_DQ);
    if (!rc) rc = pcl_get_q(s, dq, 1);
    if (!rc) rc = pcl_select(s, PCL_REG_Q);
    if (rc) layer1_release_locked();
    return rc;
}

int pcl_sharp_flux1(int rp, const double *rp_params, int lim_type, int meqn, int mwaves, int maux, int mcapa,
                    int mbc, int mx, const double *q, double *dq, const double *aux, double dx, double dt,
                    double *cfl) {
    return host_sharp(1, rp, rp_params, lim_type, meqn, mwaves, maux, mcapa, mbc, mx, 1, q, dq, aux, dx, 1.0, dt, cfl);
}

int pcl_sharp_flux2(int rp, const double *rp_params, int lim_type, int meqn, int mwaves, int maux, int mcapa,
                    int mbc, int mx, int my, const double *q, double *dq, const double *aux, double dx,
                    double dy, double dt, double *cfl) {
    return host_sharp(2, rp, rp_params, lim_type, meqn, mwaves, maux, mcapa, mbc, mx, my, q, dq, aux, dx, dy, dt, cfl);
}

// ---- multi-GPU -----------------------------------------------------------------------------------
// The halo stream gets the HIGHEST stream priority.  Two reasons: (i) what runs on it -- pack, Send/Recv, unpack, rim
// tiles -- is the critical path of a decomposed step and should win the arbitration against the interior tiles;
// (ii) the runtime multiplexes the streams of one priority class onto a few hardware queues, and two streams that land
// on the same queue run strictly one after the other (seen in a kernel trace: every launch of both streams on one
// queue id, the rim tiles behind the interior ones): streams of different priority never share a queue.
// PCL_HALO_PRIORITY=0 keeps the default priority (A/B).
static hipError_t create_halo_stream(hipStream_t *st) {
    const char *e = getenv("PCL_HALO_PRIORITY");
    if (e && atoi(e) == 0) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    int least = 0, greatest = 0;
    hipError_t rc = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (rc != hipSuccess) return rc;
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest);
}

int pcl_comm_unique_id(char uid[128]) {
    std::string err;
    if (pcl::Halo::unique_id(uid, err)) return fail(PCL_ECOMM, err);
    return PCL_OK;
}

int pcl_comm_check(int nranks, int rank, const int neighbors[8]) {
    if (!neighbors) return fail(PCL_EINVAL, "null argument");
    if (nranks < 1) return fail(PCL_EINVAL, "pcl_comm_init: nranks must be >= 1 (got " + std::to_string(nranks) + ")");
    if (rank < 0 || rank >= nranks)
        return fail(PCL_EINVAL, "pcl_comm_init: rank " + std::to_string(rank) + " outside 0.." + std::to_string(nranks - 1));
    static const char *dirs[8] = {"W", "E", "S", "N", "SW", "SE", "NW", "NE"};
    for (int d = 0; d < 8; d++) {
        const int nb = neighbors[d];
        if (nb < -1 || nb >= nranks)
            return fail(PCL_EINVAL, std::string("pcl_comm_init: neighbour ") + dirs[d] + " = " + std::to_string(nb) +
                                        " is not a rank of this communicator (0.." + std::to_string(nranks - 1) + ", or -1)");
    }
    // A block is its own neighbour only across a periodic dimension that it spans alone; then BOTH faces of that
    // dimension wrap onto it.  One face = self and the other not means the caller's neighbour table is wrong.
    if ((neighbors[0] == rank) != (neighbors[1] == rank))
        return fail(PCL_EINVAL, "pcl_comm_init: W/E neighbours: a block that wraps onto itself does so on both faces");
    if ((neighbors[2] == rank) != (neighbors[3] == rank))
        return fail(PCL_EINVAL, "pcl_comm_init: S/N neighbours: a block that wraps onto itself does so on both faces");
    return PCL_OK;
}

int pcl_comm_init(pcl_solver *s, int nranks, int rank, const char uid[128], const int neighbors[8]) {
    if (!s || !uid || !neighbors) return fail(PCL_EINVAL, "null argument");
    // argument validation BEFORE anything reaches RCCL (whose own diagnostics for these mistakes is a bare
    // "invalid usage" from ncclCommInitRank)
    if (int rc = pcl_comm_check(nranks, rank, neighbors)) return rc;
    if (nranks > 1 && getenv("PCL_FORCE_DEVICE"))
        return fail(PCL_EINVAL, "pcl_comm_init: PCL_FORCE_DEVICE pins every rank to one GPU; RCCL needs one device per rank");
    HIP_TRY(hipSetDevice(s->cfg.device));
    std::string err;
    const int nmax = s->cfg.meqn > s->cfg.maux ? s->cfg.meqn : s->cfg.maux;
    // 2-D: the (i, j) plane of cells.  3-D: blocks are cut in y and z only; the exchanged plane is (j, k) and
    // each element is a whole row of I cells (x ghost cells included: they are refilled by the physical BCs)
    const int rc3 = s->cfg.ndim == 3
                        ? s->halo.init(nranks, rank, uid, neighbors, s->J, s->K, s->cfg.mbc, nmax, s->stream, err, s->I,
                                       s->pitch, s->pitch * s->J)
                        : s->halo.init(nranks, rank, uid, neighbors, s->I, s->J, s->cfg.mbc, nmax, s->stream, err, 1, 1,
                                       s->pitch, s->cfg.ndim == 1 ? 0 : -1);     // 1-D: a single row, W / E strips only
    if (rc3) return fail(PCL_ECOMM, err);
    if (!s->hstream) {
        HIP_TRY(create_halo_stream(&s->hstream));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_h0, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_h1, hipEventDisableTiming));
    }
    const char *e = getenv("PCL_HALO_OVERLAP");
    s->overlap = e ? atoi(e) : 1;
    s->overlap_dflt = e == nullptr;
    return PCL_OK;
}

int pcl_comm_init_host(pcl_solver *s, int nranks, int rank, const int neighbors[8], pcl_host_exchange_fn xfn,
                       pcl_host_reduce_fn rfn, void *user) {
    if (!s || !neighbors || !xfn || !rfn) return fail(PCL_EINVAL, "null argument");
    if (int rc = pcl_comm_check(nranks, rank, neighbors)) return rc;
    HIP_TRY(hipSetDevice(s->cfg.device));
    std::string err;
    const int nmax = s->cfg.meqn > s->cfg.maux ? s->cfg.meqn : s->cfg.maux;
    const int rc3 = s->cfg.ndim == 3
                        ? s->halo.init_host(nranks, rank, neighbors, s->J, s->K, s->cfg.mbc, nmax, s->stream, xfn, rfn, user,
                                            err, s->I, s->pitch, s->pitch * s->J)
                        : s->halo.init_host(nranks, rank, neighbors, s->I, s->J, s->cfg.mbc, nmax, s->stream, xfn, rfn, user,
                                            err, 1, 1, s->pitch, s->cfg.ndim == 1 ? 0 : -1);
    if (rc3) return fail(PCL_ECOMM, err);
    if (!s->hstream) {
        HIP_TRY(create_halo_stream(&s->hstream));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_h0, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s->ev_h1, hipEventDisableTiming));
    }
    const char *e = getenv("PCL_HALO_OVERLAP");
    s->overlap = e ? atoi(e) : 1;
    s->overlap_dflt = e == nullptr;
    return PCL_OK;
}

int pcl_halo_can_overlap(pcl_solver *s, int *yes) {
    if (!s || !yes) return fail(PCL_EINVAL, "null argument");
    int box[4], ntiles[2];
    const bool base = s->halo.active && s->cfg.kind == PCL_KIND_CLASSIC && s->cfg.ndim == 2 && s->cfg.method[2] < 0 &&
                      s->cfg.meqn <= 8 && s->overlap == 1;
    // 2: this block can also run the one-kernel step with interior / rim tile subsets; 1: the two-pass step only
    *yes = !base ? 0
           : (fused_step_ok(s) && pcl::exact::step2ds_interior_box(make_args(s, s->q, s->t2, 1, 1.0), box, ntiles)) ? 2
           : (twopass_overlap_ok(s) && pcl::exact::x_interior_box(make_args(s, s->q, s->t1, 1, 1.0), box, ntiles)) ? 1 : 0;
    return PCL_OK;
}

int pcl_halo_exchange_ahead(pcl_solver *s, int on) {
    if (!s) return fail(PCL_EINVAL, "null argument");
    if (on) {
        int yes = 0;
        if (int rc = pcl_halo_can_overlap(s, &yes)) return rc;
        if (!yes) return fail(PCL_ESTATE, "pcl_halo_exchange_ahead: needs the overlapped dimension-split 2-D step of a decomposed run "
                                          "(pcl_comm_init done, PCL_HALO_OVERLAP=1, a block with interior x-pass tiles)");
        if (on == 2 && yes != 2)
            return fail(PCL_ESTATE, "pcl_halo_exchange_ahead(2): this block has no interior box of one-kernel tiles (pcl_halo_can_overlap < 2)");
        if (!s->ev_y) HIP_TRY(hipEventCreateWithFlags(&s->ev_y, hipEventDisableTiming));
    }
    // 1: the two-pass step's order on the communicator (all-reduce, then the new state's exchange); 2: the one-kernel
    // step's (the exchange behind the rim tiles, then the all-reduce).  Every rank of a run must hold the same value.
    s->exchange_ahead = on == 2 ? 2 : on ? 1 : 0;
    s->ghosts_drop_all();
    return PCL_OK;
}

int pcl_halo_exchange(pcl_solver *s) {
    if (s) s->ghosts_drop_all();          // exchange-ahead: whatever filled the ghost frames no longer holds
    if (!s) return fail(PCL_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    std::string err;
    if (s->halo.exchange(cur(s), s->cfg.meqn, s->pitch, s->plane, err)) return fail(PCL_ECOMM, err);
    return PCL_OK;
}

int pcl_halo_exchange_aux(pcl_solver *s) {
    if (!s) return fail(PCL_EINVAL, "null argument");
    if (s->cfg.maux <= 0) return PCL_OK;
    HIP_TRY(hipSetDevice(s->cfg.device));
    std::string err;
    if (s->halo.exchange(s->aux, s->cfg.maux, s->pitch, s->plane, err)) return fail(PCL_ECOMM, err);
    return PCL_OK;
}

int pcl_halo_region(int dir, int send, int I, int J, int mbc, int out[4]) {
    if (dir < 0 || dir > 7 || !out || mbc < 1 || I <= 2 * mbc || J <= 2 * mbc) return fail(PCL_EINVAL, "bad argument");
    const pcl::HaloRegion r = pcl::halo_region(dir, send != 0, I, J, mbc);
    out[0] = r.i0; out[1] = r.j0; out[2] = r.ni; out[3] = r.nj;
    return PCL_OK;
}

int pcl_allreduce_max(pcl_solver *s, double *value) {
    if (!s || !value) return fail(PCL_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(s->cfg.device));
    std::string err;
    if (s->halo.allreduce_max(value, err)) return fail(PCL_ECOMM, err);
    return PCL_OK;
}

}  // extern "C"
