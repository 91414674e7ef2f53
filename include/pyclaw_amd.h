/*
 * pyclaw_amd.h -- C ABI of libpyclaw_amd.so (MI355X / gfx950 HIP implementation of
 * PyClaw's classic + SharpClaw hot path).
 *
 * What this boundary replaces in the reference (paths relative to the reference tree):
 * the f2py extension modules classic1/classic2/(classic3)/sharpclaw1/sharpclaw2 that
 * src/pyclaw/clawpack.py:323,538-552 and src/pyclaw/sharpclaw.py:385,558 import and
 * call, plus the ghost-cell/backup array traffic those Python drivers do around them
 * (src/pyclaw/solver.py:315-452, :660, :690; src/petclaw/state.py:236-269 for the
 * halo exchange and src/petclaw/cfl.py:29-31 for the CFL all-reduce).
 *
 * Two layers:
 *   1. "f2py-shaped" stateless calls on HOST arrays (pcl_step1 / pcl_step2ds /
 *      pcl_step2): same argument meaning as the Fortran subroutines, Fortran array
 *      order, caller owns every array.  A maintainer can bind these in place of the
 *      f2py modules without touching the Python drivers (PCIe traffic every call).
 *   2. A resident solver handle (pcl_create ...): q/aux live in HBM across steps,
 *      ghost cells, backup/restore, CFL reduction and halo exchange run on the device.
 *      This is what the pyclaw_amd ClawSolver / SharpClawSolver classes use.
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative
 * PCL_E* code and leaves a message for pcl_last_error(); host arrays are Fortran
 * ordered float64 with the component index fastest, q(m,i,j) at
 * q[(m) + meqn*((i) + (mx+2*mbc)*(j))] (0-based, ghost cells included) exactly like the
 * reference's qbc (doc/differences.rst:9-17).  One host thread per solver handle.
 * There is NO CPU fallback: every entry point fails with PCL_ENODEVICE without a GPU.
 */
#ifndef PYCLAW_AMD_H
#define PYCLAW_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define PCL_MAX_WAVES 8
#define PCL_MAX_RP_PARAMS 8

/* error codes */
#define PCL_OK 0
#define PCL_EINVAL (-1)    /* bad argument / unsupported configuration           */
#define PCL_ENODEVICE (-2) /* no usable HIP device                               */
#define PCL_EHIP (-3)      /* a HIP runtime call failed                          */
#define PCL_ECOMM (-4)     /* RCCL failure                                       */
#define PCL_ESTATE (-5)    /* call not valid in the handle's current state       */

/* Riemann solver ids.  The reference selects the solver at link time (the rpn2/rpt2
 * symbols of the app Makefile, e.g. apps/euler/2d/shockbubble/Makefile:3) and passes
 * its scalars through `common /cparam/` (src/pyclaw/state.py:142-162); here it is an
 * id plus the cparam values in declaration order. */
#define PCL_RP_ADVECTION_1D 1 /* rp1_advection.f            cparam: u                   */
#define PCL_RP_ACOUSTICS_1D 2 /* rp1_acoustics.f            cparam: rho,bulk,cc,zz      */
#define PCL_RP_BURGERS_1D 3     /* rp1_burgers.f90 (transonic entropy fix); no cparam            */
#define PCL_RP_EULER_1D 4       /* rp1_euler_with_efix.f          cparam: gamma,gamma1            */
#define PCL_RP_SHALLOW_1D 5     /* rp1_shallow_roe_with_efix.f    cparam: g                       */
#define PCL_RP_ADVECTION_COLOR_1D 6 /* rp1_advection_color.f; aux(1) = velocity at the cell's left edge */
#define PCL_RP_ELASTICITY_FWAVE_1D 7 /* rp1_nonlinear_elasticity_fwave.f (stegoton); F-WAVE solver: fwave = 1;  */
                                     /* aux(1)=rho, aux(2)=K, aux(3)=1: sigma=K eps, else exp(K eps)-1          */
#define PCL_RP_ACOUSTICS_2D 10 /* rpn2/rpt2_acoustics.f     cparam: rho,bulk,cc,zz      */
#define PCL_RP_EULER5_2D 11    /* rpn2/rpt2_euler_5wave.f   cparam: gamma,gamma1        */
#define PCL_RP_ADVECTION_2D 12  /* rpn2/rpt2_advection.f      cparam: u,v                  */
#define PCL_RP_SHALLOW_2D 13    /* rpn2/rpt2_shallow_roe_with_efix.f  cparam: g            */
#define PCL_RP_VC_ACOUSTICS_2D 14 /* rpn2/rpt2_vc_acoustics.f; aux(1)=Z, aux(2)=c */
#define PCL_RP_VC_ADVECTION_2D 15 /* rpn2/rpt2_vc_advection.f; aux(1)=u at the left edge, aux(2)=v at the bottom edge */
#define PCL_RP_SHALLOW_SPHERE_2D 16 /* rpn2/rpt2_shallow_sphere.f (apps/shallow-sphere/Makefile:7); common /sw/ g + comxyt dxcom,dycom */
                                    /* -> rp_params g, dx, dy; aux = the 16 components of apps/shallow-sphere/setaux.f:10-25;      */
                                    /* the unsplit step follows the app's step2qcor.f (conservation fix qcor.f) instead of step2.f */
#define PCL_RP_PSYSTEM_FWAVE_2D 17 /* rpn2/rpt2_psystem.f (test/psystem/Makefile:5); F-WAVE solver: fwave = 1; aux as above + aux(4)=eps */
#define PCL_RP_VC_ACOUSTICS_3D 20 /* rpn3_vc_acoustics.f (test/acoustics/3d/Makefile); aux(1)=Z, aux(2)=c; dim-split only */

/* boundary condition types = pyclaw.BC (src/pyclaw/solver.py:17-23) */
#define PCL_BC_CUSTOM 0
#define PCL_BC_OUTFLOW 1
#define PCL_BC_PERIODIC 2
#define PCL_BC_REFLECTING 3
/* not a pyclaw.BC value: the sphere app's custom y boundary (qbc_lower_y / qbc_upper_y,
 * apps/shallow-sphere/shallow_4_Rossby_Haurwitz_wave.py:295-313): ghost row j mirrors interior row 2*mbc-1-j with the
 * x index reversed over the whole ghosted width.  Accepted by pcl_bc for idim = 1. */
#define PCL_BC_SPHERE_MIRROR 4

/* arithmetic mode */
#define PCL_MATH_EXACT 0 /* no FMA contraction, IEEE divide/sqrt: bit-identical to the  */
                         /* reference Fortran built without FMA (the parity mode)        */
#define PCL_MATH_FAST 1  /* FMA contraction + reciprocal-multiply division (each quotient  */
                         /* within ~1 ulp instead of correctly rounded).  Not bit-identical;*/
                         /* tested against the reference goldens at rtol 1e-12.            */
#define PCL_MATH_STRICT 2 /* PCL_MATH_EXACT whose quotients take the IEEE division also when the    */
                          /* numerator lies in the underflow range (|n| < 2^-960: a momentum of      */
                          /* 1e-310 ahead of a front), where EXACT's shared-reciprocal quotient can  */
                          /* be one step of the denormal grid off (measured <= 3.2e-322).  3-5 %     */
                          /* slower than EXACT on compute-bound states, equal on the headline.       */

/* solver kind: classic wave propagation (clawpack.py) or SharpClaw method of lines (sharpclaw.py) */
#define PCL_KIND_CLASSIC 0
#define PCL_KIND_SHARPCLAW 1

/* SharpClaw registers: state q, the two Runge-Kutta stage registers (Solver._rk_stages,
 * solver.py:266-291), the increment dq, and one scratch register (dq_src contributions) */
#define PCL_REG_Q 0
#define PCL_REG_S1 1
#define PCL_REG_S2 2
#define PCL_REG_DQ 3
#define PCL_REG_TMP 4

typedef struct pcl_solver pcl_solver;

typedef struct pcl_config {
    int ndim;                       /* 1 or 2                                              */
    int n[3];                       /* interior cells of THIS process' block: mx,my,(mz)   */
    int mbc;                        /* ghost width (classic: 2)                            */
    int meqn, mwaves, maux;
    int method[7];                  /* as ClawSolver.set_method (clawpack.py:192-212):     */
                                    /* [1]=order, [2]=-1 dim-split | order_trans,          */
                                    /* [5]=mcapa+1 (0 = no capa), [6]=maux                 */
    int mthlim[PCL_MAX_WAVES];      /* limiter per wave family (philim.f ids 0..5)         */
    int fwave;                      /* 1: f-wave form (flux2fw.f:151-152)                  */
    int rp;                         /* PCL_RP_*                                            */
    double rp_params[PCL_MAX_RP_PARAMS];
    double d[3];                    /* dx,dy,(dz)                                          */
    int device;                     /* HIP device ordinal                                  */
    int math;                       /* PCL_MATH_*                                          */
    int kind;                       /* PCL_KIND_CLASSIC | PCL_KIND_SHARPCLAW               */
    int lim_type;                   /* SharpClaw: 1 = tvd2 (reconstruct.f90:568-625; mthlim per COMPONENT),     */
                                    /*            2 = PyWENO form (weno.f90), order 2*mbc-1: mbc = 3..9 is      */
                                    /*                weno_order 5..17 (sharpclaw.py:479), 3 = legacy WENO5     */
} pcl_config;

/* ---- library ---------------------------------------------------------------------- */
const char *pcl_last_error(void);
int pcl_version(void);
int pcl_device_count(void);                     /* never initialises a context          */

/* ---- layer 1: f2py-shaped stateless calls on host arrays ---------------------------- */
/* Stateless for the caller like the f2py modules; internally the device buffers of the last call are kept and reused
 * while the array shapes stay the same.  pcl_layer1_release() frees them (optional: also done at unload). */
void pcl_layer1_release(void);
/* Arithmetic mode of the layer-1 calls that follow (default PCL_MATH_EXACT, the f2py modules' results bit for bit). */
int pcl_layer1_math(int math);
/* classic1.step1(mbc,mx,q,aux,dx,dt,method,mthlim) -> (q,cfl)   step1.f:4-5, clawpack.py:323.
 * q(meqn,1-mbc:mx+mbc) is updated in place for cells 1..mx (the two ghost cells the
 * Fortran also touches are left unchanged: no caller reads them, clawpack.py:406). */
int pcl_step1(int rp, const double *rp_params, int meqn, int mwaves, int maux, int mbc, int mx,
              double *q, const double *aux, double dx, double dt, const int *method,
              const int *mthlim, double *cfl);

/* classic1fw.step1(...) -> (q,cfl): the f-wave twin, src/fortran/1d/classic/step1fw.f (clawpack.py:221-222 picks the
 * module name + 'fw' when solver.fwave is set); rp must be an f-wave solver. */
int pcl_step1fw(int rp, const double *rp_params, int meqn, int mwaves, int maux, int mbc, int mx,
                double *q, const double *aux, double dx, double dt, const int *method,
                const int *mthlim, double *cfl);

/* classic2.step2ds(maxm,mbc,mx,my,qold,qnew,aux,dx,dy,dt,method,mthlim,aux1,aux2,aux3,work,ids)
 * -> (qnew,cfl)   step2ds.f:2-5, clawpack.py:538-544.  qold may alias qnew.  The work
 * arrays of the Fortran signature have no meaning here and are not taken. */
int pcl_step2ds(int rp, const double *rp_params, int fwave, int meqn, int mwaves, int maux,
                int mbc, int mx, int my, const double *qold, double *qnew, const double *aux,
                double dx, double dy, double dt, const int *method, const int *mthlim,
                double *cfl, int ids);

/* classic2.step2(...) -> (qnew,cfl)   step2.f:2-5, clawpack.py:550-552 (unsplit). */
int pcl_step2(int rp, const double *rp_params, int fwave, int meqn, int mwaves, int maux,
              int mbc, int mx, int my, const double *qold, double *qnew, const double *aux,
              double dx, double dy, double dt, const int *method, const int *mthlim,
              double *cfl);

/* classic3.step3ds(maxm,mbc,mx,my,mz,qold,qnew,aux,dx,dy,dz,dt,method,mthlim,aux1,aux2,aux3,work,idir)
 * -> (qnew,cfl)   src/fortran/3d/classic/step3ds.f:2-5, clawpack.py:678-688: one directional sweep
 * (idir 1..3) of the dimension-split 3-D algorithm; q(m,i,j,k) component fastest, ghost cells included;
 * slices with transverse indices 0..m+1 are swept, everything else is returned unchanged. */
int pcl_step3ds(int rp, const double *rp_params, int meqn, int mwaves, int maux, int mbc, int mx, int my,
                int mz, const double *qold, double *qnew, const double *aux, double dx, double dy, double dz,
                double dt, const int *method, const int *mthlim, double *cfl, int idir);

/* classic3.step3(maxm,mbc,mx,my,mz,qold,qnew,aux,dx,dy,dz,dt,method,mthlim,aux1,aux2,aux3,work) -> (qnew,cfl)
 * src/fortran/3d/classic/step3.f:2-6, clawpack.py:690-696: the UNSPLIT 3-D step with the transverse solves rpt3 / rptt3
 * (flux3.f:260-593); method[2] = order_trans = 0, 10, 11, 20, 21 or 22.  Interior cells of qnew are updated, ghost
 * cells returned unchanged. */
int pcl_step3(int rp, const double *rp_params, int meqn, int mwaves, int maux, int mbc, int mx, int my, int mz,
              const double *qold, double *qnew, const double *aux, double dx, double dy, double dz, double dt,
              const int *method, const int *mthlim, double *cfl);

/* sharpclaw1.flux1(q,aux,dt,t,ixy,mx,mbc,maxnx) -> (dq,cfl)   1d/sharpclaw/flux1.f90, sharpclaw.py:385
 * sharpclaw2.flux2(q,aux,dt,t,mbc,maxm,mx,my)   -> (dq,cfl)   2d/sharpclaw/flux2.f90:2, sharpclaw.py:558
 * q and dq are (meqn, mx+2mbc[, my+2mbc]) with mbc = (weno_order+1)/2 = 3..9 (clawparams.weno_order is
 * implied by mbc: lim_type 2 with mbc = k runs weno(2k-1); lim_type 1 and 3 take mbc = 3); dq's interior receives
 * dt*dq/dt, its ghost cells are zeroed.  The F90 module state the reference sets through
 * clawparams/workspace/reconstruct (lim_type, mcapa, dx, mwaves) is passed explicitly. */
/* clawparams.mthlim, part of the F90 module state the reference sets before calling flux1/flux2
 * (sharpclaw.py:268); read by lim_type = 1 (tvd2) only, indexed by component.  Default: all 1 (minmod). */
int pcl_sharp_module_mthlim(const int *mthlim, int n);
/* clawparams.char_decomp of the same module state (sharpclaw.py:262), read by pcl_sharp_flux1: 0 = component-wise
 * reconstruction, 1 = wave-based (1d/sharpclaw/flux1.f90:80-107: rp1 on the cell averages, then tvd2_wave for
 * lim_type 1 -- mthlim indexed by WAVE there -- or weno5_wave for lim_type 2; reconstruct.f90:393-478,728-806).
 * 2 and 3 need a user-supplied evec routine the reference only stubs (evec.f90:13-14), and the 2-D flux1.f90 calls
 * rpn2 with a wrong argument list on the char_decomp = 1 path (2d/sharpclaw/flux1.f90:86): neither can run in the
 * reference, neither is offered.  A resident solver takes the value in cfg.method[4] (method(5), unused by SharpClaw). */
int pcl_sharp_module_char_decomp(int char_decomp);
int pcl_sharp_flux1(int rp, const double *rp_params, int lim_type, int meqn, int mwaves, int maux, int mcapa,
                    int mbc, int mx, const double *q, double *dq, const double *aux, double dx, double dt,
                    double *cfl);
int pcl_sharp_flux2(int rp, const double *rp_params, int lim_type, int meqn, int mwaves, int maux, int mcapa,
                    int mbc, int mx, int my, const double *q, double *dq, const double *aux, double dx,
                    double dy, double dt, double *cfl);

/* ---- layer 2: resident solver -------------------------------------------------------- */
int pcl_create(const pcl_config *cfg, pcl_solver **out);
void pcl_destroy(pcl_solver *s);

/* Host <-> device copies of the whole block.  with_ghosts=0: host array is state.q,
 * (meqn,mx[,my]) Fortran order (get_qbc_from_q / set_q_from_qbc, state.py:171-206);
 * with_ghosts=1: host array is qbc, (meqn,mx+2mbc[,my+2mbc]). */
int pcl_put_q(pcl_solver *s, const double *host, int with_ghosts);
int pcl_get_q(pcl_solver *s, double *host, int with_ghosts);
int pcl_put_aux(pcl_solver *s, const double *host_auxbc); /* always with ghosts (auxbc)   */

/* Ghost fill of one side of one dimension on the device: solver.py:384-452.
 * side 0 = lower, 1 = upper.  PCL_BC_CUSTOM is not accepted here: use the strip calls
 * (arbitrary user numpy code) or pcl_bc_const (inflow of a constant state). */
int pcl_bc(pcl_solver *s, int idim, int side, int bctype);
int pcl_bc_const(pcl_solver *s, int idim, int side, const double *state /* [meqn] */);
/* Same fill for the aux array (auxbc_lower/upper, solver.py:526-596: reflecting does not negate). */
int pcl_bc_aux(pcl_solver *s, int idim, int side, int bctype);
/* Copy the `width` outermost layers (ghost cells first) of one side to/from a host
 * array shaped like qbc with that dimension cut to `width`: for custom BCs written in
 * Python that only touch the strip (user_bc_lower/upper, solver.py:404-405,439-440).  1-D, 2-D and 3-D. */
int pcl_get_strip(pcl_solver *s, int idim, int side, int width, double *host);
int pcl_put_strip(pcl_solver *s, int idim, int side, int width, const double *host);
/* the aux twin of pcl_put_strip: a ghost strip of auxbc filled by a Python aux-BC callback (user_aux_bc_lower/upper,
 * solver.py:526-596), placed after the aux halo exchange of a decomposed run */
int pcl_put_aux_strip(pcl_solver *s, int idim, int side, int width, const double *host);
/* Gauges (Solver.write_gauge_values, solver.py:731-741: q[:,x,y] and aux[:,x,y] of a few cells
 * after every step): gather `ncell` interior cells (0-based interior indices ij[2*c], ij[2*c+1];
 * the second is ignored in 1-D; on a 3-D grid the tuples are triples ij[3*c], ij[3*c+1], ij[3*c+2]) of the resident q -- and of aux when `aux` is not NULL -- into
 * q[ncell][meqn] / aux[ncell][maux].  One small kernel + one D2H of ncell*(meqn+maux) doubles
 * instead of reading the whole state back every step. */
int pcl_get_cells(pcl_solver *s, int ncell, const int *ij, double *q, double *aux);

/* One homogeneous step of length dt on the resident state (ghost cells must be filled):
 * dim-split  = step2ds(ids=1) then step2ds(ids=2)          clawpack.py:538-546
 * unsplit    = step2                                       clawpack.py:550-552
 * 1-D        = step1                                       clawpack.py:323
 * *cfl receives the max Courant number -- over all blocks once pcl_comm_init has been called (the
 * all-reduce of petclaw/cfl.py:29-31 runs on the device before the read-back).  The pre-step state
 * stays available until the next call: pcl_undo_step() makes it current again (the
 * reference's q_backup / retake path, solver.py:660,690) without any copy. */
int pcl_step_hyperbolic(pcl_solver *s, double dt, double *cfl);
int pcl_undo_step(pcl_solver *s);
/* apply_q_bcs + step_hyperbolic in ONE call (clawpack.py:528-555 with solver.py:354-381): the ghost
 * fills of the listed sides, in the reference's order (dimension by dimension, lower then upper),
 * then the step.  bc[2*idim+side] = PCL_BC_OUTFLOW/PERIODIC/REFLECTING, PCL_BC_CUSTOM with
 * cstate[(2*idim+side)*PCL_MAX_RP_PARAMS ...] = constant state, or -1 = leave that side alone (not at a
 * physical boundary / periodic handled by the halo).  Does the halo exchange first when pcl_comm_init
 * has been called. */
int pcl_bc_step(pcl_solver *s, const int *bc, const double *cstate, double dt, double *cfl);
/* Single directional sweep (ids = 1 or 2), qold -> qnew like step2ds; used by layer 1. */
int pcl_sweep(pcl_solver *s, int ids, double dt, double *cfl);

/* Explicit device-side backup/restore (needed when user code modifies q before the
 * hyperbolic step: Strang source splitting, start_step). */
int pcl_backup(pcl_solver *s);
int pcl_restore(pcl_solver *s);

/* Built-in device source terms for the reference apps' step_src callbacks.
 * id 1: Euler radial-symmetry source, 2-stage RK (test/euler/2d/shockbubble.py:59-94);
 *       aux[0] = radial coordinate; params = {gamma1, ndim}. */
#define PCL_SRC_EULER_RADIAL 1
/* id 2: Coriolis force of the shallow-water-on-the-sphere app, apps/shallow-sphere/src2.f:43-146 (projection onto the
 *       tangent plane, 4-stage RK, projection); aux components 14-16 = radial unit vector; no params. */
#define PCL_SRC_SPHERE_CORIOLIS 2
int pcl_src(pcl_solver *s, int src_id, double dt, const double *params, int nparams);
/* Godunov-split source term (clawpack.py:156-159: step_src(dt) after an accepted hyperbolic step) applied by the LAST
 * pass of the 2-D step (y pass of step2ds, y phase of step2) while it stores its results, instead of one more read + write of q by
 * pcl_src: same arithmetic, same bits.  src_id = PCL_SRC_EULER_RADIAL with params {gamma1, ndim}, or 0 to switch it
 * off.  A rejected step discards the pass' output with the rest (pcl_undo_step), as step() returns before the source. */
int pcl_fuse_source(pcl_solver *s, int src_id, const double *params, int nparams);

/* ---- SharpClaw (kind = PCL_KIND_SHARPCLAW) ----------------------------------------------------- */
/* Which register the put/get/bc/strip/halo calls act on (default PCL_REG_Q): the RK stages get
 * their ghost cells filled exactly like q (apply_q_bcs(stage), sharpclaw.py:546). */
int pcl_select(pcl_solver *s, int reg);
/* deltaq += dq_src(stage) of sharpclaw.py:232-235 evaluated by the LAST pass of every stage while it stores deltaq (or
 * its RK combination), for the built-in twin of the shock-bubble app's dq_Euler_radial
 * (apps/euler/2d/shockbubble/shockbubble.py:95-122): src_id = PCL_SRC_EULER_RADIAL with params {gamma1, ndim}; 0 switches
 * it off.  Needs euler_5wave_2d, lim_type 2 with mbc 3 (WENO5), no capacity function, aux(1) = radial coordinate. */
int pcl_sharp_fuse_dq_src(pcl_solver *s, int src_id, const double *params, int nparams);

/* sharpclaw1.flux1 / sharpclaw2.flux2 (sharpclaw.py:385,558; flux1.f90, flux2.f90): dq register :=
 * dt * dq/dt of the selected register (ghost cells must be filled); *cfl = max Courant number (global
 * after pcl_comm_init). */
int pcl_sharp_dq(pcl_solver *s, double dt, double *cfl);
/* One Runge-Kutta stage in two kernels (2-D): dq of the selected register (pcl_sharp_dq) with the
 * combination that consumes it fused into the last directional pass, so deltaq is never written:
 *   op 1: D = A + dq/ca      op 2: D = ca*A + cb*(B + dq)      op 5: D = A + cb*B + cc*dq
 * (the expressions of SharpClawSolver.step, sharpclaw.py:168-206, in the numpy evaluation order).
 * The stage's Courant number (all-reduced over the blocks in a decomposed run) is returned in *cfl; the
 * result becomes register D only if cfl <= cfl_max (dq() raises CFLError before the update otherwise,
 * sharpclaw.py:228-230).  D, A, B in {PCL_REG_Q, PCL_REG_S1, PCL_REG_S2}; D's ghost cells are left
 * unset (the next dq fills them, like the reference's apply_q_bcs on each stage). */
int pcl_sharp_stage(pcl_solver *s, double dt, int op, int D, int A, int B, double ca, double cb, double cc,
                    double cfl_max, double *cfl);

/* pcl_sharp_dq / pcl_sharp_stage preceded by the stage's apply_q_bcs (sharpclaw.py:347, solver.py:354-381) in the
 * same call: halo exchange of the selected register in a decomposed run, then the physical BCs (`bc`, `cstate` as
 * for pcl_bc_step: 2*ndim types, <0 = no fill on that side, PCL_BC_CUSTOM = constant state).  In a decomposed 2-D
 * run the ghost frame is built on the halo stream while the x pass runs the tiles that read no ghost cell
 * (PCL_HALO_OVERLAP=0 keeps everything on one stream). */
int pcl_sharp_bc_dq(pcl_solver *s, const int *bc, const double *cstate, double dt, double *cfl);
int pcl_sharp_bc_stage(pcl_solver *s, const int *bc, const double *cstate, double dt, int op, int D, int A, int B,
                       double ca, double cb, double cc, double cfl_max, double *cfl);

/* Register arithmetic of the Runge-Kutta schemes (sharpclaw.py:168-206), evaluated in the order
 * written: op 1: D = A + B/ca   2: D = ca*A + cb*(B + C)   3: D = A/ca + cb*B
 *          4: D = ca*A - cb*B   5: D = A + cb*B + cc*C
 *          6: C = A/ca + cb*B, then D = cc*C - 5*B with D == B (ops 3 and 4 of SSP104's mid-step, sharpclaw.py:186-187,
 *             in one pass over the registers).  D, A, B, C are PCL_REG_* ids. */
int pcl_rk_op(pcl_solver *s, int op, int D, int A, int B, int Cc, double ca, double cb, double cc);

int pcl_sync(pcl_solver *s);

/* Timing hooks for bench.py: HIP events recorded on the solver's compute stream. */
int pcl_timer_start(pcl_solver *s);
int pcl_timer_stop(pcl_solver *s, float *ms);
/* Cumulative device time (ms) and launch count of the sweep kernels since the last reset, from HIP events around
 * the sweep launches (enable before use).  enable = 0: off; 1: every step; N > 1: every N-th step (the event
 * records sit between the kernels and cost a few microseconds of dispatch gap each; sampling keeps the average
 * launch duration and removes most of that). */
int pcl_kernel_timing(pcl_solver *s, int enable);
int pcl_kernel_timing_read(pcl_solver *s, double *ms_total, long *launches);
/* The dimension-split 2-D step has two forms with identical results (step2ds.f:83-159 as x pass + y pass, or both sweeps
 * in one kernel, q through HBM once per step); one block of an aux-free solver runs the faster one, re-measured every 256
 * steps (PCL_TUNE_FUSED_STEP = 0 / 1 pins a form).  Since the last pcl_kernel_timing reset: cumulative device time (ms)
 * and sampled launch count of the one-kernel step (pcl_kernel_timing_read holds the two passes), and the steps each
 * form has run. */
int pcl_step_form_stats(pcl_solver *s, double *ms_total, long *launches, long *steps_one_kernel, long *steps_two_pass);
/* hyperbolic steps (classic) / right-hand sides (SharpClaw) attempted since pcl_create, rejected ones included */
int pcl_step_count(pcl_solver *s, long *steps);

/* ---- multi-GPU: one block per process, RCCL over xGMI -------------------------------- */
/* Replaces PETSc DMDA globalToLocal (src/petclaw/state.py:254-269) and Vec.max
 * (src/petclaw/cfl.py:29-31).  uid is the 128-byte ncclUniqueId produced by
 * pcl_comm_unique_id() on rank 0 and distributed by the caller (any side channel).
 * neighbors[8]: rank of the W,E,S,N,SW,SE,NW,NE neighbour block or -1 (physical
 * boundary without periodic wrap). */
int pcl_comm_unique_id(char uid[128]);
int pcl_comm_init(pcl_solver *s, int nranks, int rank, const char uid[128],
                  const int neighbors[8]);
/* Exchange-ahead for the overlapped dimension-split 2-D step (pcl_bc_step): with on = 1 the halo exchange of the NEW
 * state is enqueued on the halo stream right behind the y pass that produced it -- it travels while the Courant number
 * is handed over, the host decides accept / retake and the next x pass starts its interior tiles -- and the next
 * pcl_bc_step skips its own exchange when nothing has touched the state in between (any other call that changes q
 * or its ghost cells makes it exchange again; a rejected step goes back to the pre-step buffer, whose ghost frame is
 * still filled).  petclaw has no counterpart (globalToLocal is blocking, petclaw/state.py:254-262).  EVERY rank of
 * the communicator must make the same choice (the order of operations on the communicator depends on it):
 * pcl_halo_can_overlap tells whether this rank's block qualifies (decomposed, dim-split 2-D, interior x-pass tiles,
 * PCL_HALO_OVERLAP=1); the caller agrees over all ranks (pyclaw_amd/clawpack.py) and then switches it on everywhere.
 * pcl_halo_can_overlap: 0 no, 1 yes, 2 yes and the block can also run the one-kernel form of the step (both sweeps in
 * one launch) with interior / rim tile subsets.  That form sends the new state's halo behind its RIM tiles, beside the
 * interior launch and BEFORE the Courant number's all-reduce: on = 2 selects that order and is only valid when every
 * rank reported 2; on = 1 keeps the two-pass order (all-reduce, then the exchange) on every rank. */
int pcl_halo_can_overlap(pcl_solver *s, int *yes);
int pcl_halo_exchange_ahead(pcl_solver *s, int on);
/* The same decomposed device path with a host-staged wire instead of RCCL (diagnostics, and multi-process tests on
 * ONE device: RCCL refuses two ranks per GPU).  Packed strips go to pinned host memory, `xfn` carries them between
 * the processes, the received strips go back up; the CFL maximum goes through `rfn`.  send/recv: all 8 directions'
 * strips back to back, direction d at doubles [off[d]*nm, (off[d]+cnt[d])*nm); what is SENT towards d arrives in the
 * receiver's region opposite(d).  Both callbacks return 0 on success and are called by every rank of the group at
 * the same points (collectives).  No reference counterpart. */
typedef int (*pcl_host_exchange_fn)(void *user, const double *send, double *recv, const long *off, const long *cnt,
                                    const int *nbr, int nm);
typedef int (*pcl_host_reduce_fn)(void *user, double *value);     /* in-place max over the ranks */
int pcl_comm_init_host(pcl_solver *s, int nranks, int rank, const int neighbors[8], pcl_host_exchange_fn xfn,
                       pcl_host_reduce_fn rfn, void *user);
/* Host-only argument check of pcl_comm_init (no GPU, no RCCL): rank inside 0..nranks-1, every neighbour a
 * rank of the communicator or -1, a self-neighbour only on both faces of a dimension.  pcl_comm_init runs it
 * first, so misuse fails with a message instead of RCCL's "invalid usage". */
int pcl_comm_check(int nranks, int rank, const int neighbors[8]);
int pcl_halo_exchange(pcl_solver *s);          /* faces + corners, width mbc            */
int pcl_halo_exchange_aux(pcl_solver *s);
int pcl_allreduce_max(pcl_solver *s, double *value);
/* Host-only helper (no GPU needed): the cell window [i0,i0+ni) x [j0,j0+nj) of a block of
 * I x J cells (ghosts included, width mbc) that is SENT towards direction dir (send=1: interior
 * cells next to that edge/corner) or FILLED from direction dir (send=0: ghost cells).  dir:
 * 0..7 = W,E,S,N,SW,SE,NW,NE.  Exactly the geometry the device pack/unpack kernels use. */
int pcl_halo_region(int dir, int send, int I, int J, int mbc, int out_i0_j0_ni_nj[4]);

/* ---- debug / self-test --------------------------------------------------------------- */
/* Runs the wavefront neighbour-shift primitive on 64 values: left[l]=in[l-1],
 * right[l]=in[l+1] (the end lane without a source reads 0). */
int pcl_debug_wave_shift(const double *in64, double *left64, double *right64);

#ifdef __cplusplus
}
#endif
#endif /* PYCLAW_AMD_H */
