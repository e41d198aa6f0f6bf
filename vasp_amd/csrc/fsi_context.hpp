// Device-resident context of one monolithic FSI problem (one per GPU).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/vaspfsi.h"
#include "fsi_element.hpp"

namespace fsi {
struct BcrData;               // exact solve of the solid cycle's coarse level by block cyclic reduction (fsi_bcr.hip)

constexpr int NQ = 24;        // Keast degree-6 rule
constexpr int NLOC = 64;      // local dofs per tet: 30 (d) + 30 (v) + 4 (p)
constexpr int MAX_REGIONS = 8;

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return hipSuccess;
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e != hipSuccess) return e;
    // Fresh device memory is whatever the previous owner left (another context of the same process included): every buffer
    // starts from zeros, so that no result can depend on it.  FSI_DEBUG_POISON=1 fills the floating-point buffers with NaN
    // bit patterns instead - a read before the first write then shows up in the first norm that is taken.
    static const bool poison = getenv("FSI_DEBUG_POISON") != nullptr;
    return hipMemset(p, (poison && std::is_floating_point<T>::value) ? 0xFF : 0, count * sizeof(T));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

// One colour of the multicolour ordering: `ngroups` groups of `group_rows` consecutive rows starting at `first_row`
// (6 rows per P2 node for the d/v block, 1 row per vertex for the pressure block).
struct Level {
  int64_t first_row, ngroups;
  int group_rows;
};

// A square sparse field block in CSR on structure arrays owned by the context.
struct SubMat {
  int64_t n = 0, nnz = 0;
  const int64_t* rowptr = nullptr;
  const int32_t* cols = nullptr;
  const int64_t* diagpos = nullptr;
  DevBuf<double> vals;
};

// Phase timer on the solver stream.  Event pairs are recorded into a ring and resolved lazily (when the ring wraps or
// when fsi_get_timers reads the totals), so that bracketing a phase never makes the host wait for the device.
struct PhaseTimer {
  static constexpr int RING = 32;
  double ms = 0.0;
  int64_t calls = 0;
  hipEvent_t e0[RING] = {}, e1[RING] = {};
  int64_t issued = 0, resolved = 0;          // pairs recorded / pairs already added to ms
};

}  // namespace fsi

struct FsiCtx {
  FsiTuning tune{};                          // what the context was created with (fsi_create_tuned); the fields below are set from it
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // sizes
  int64_t V = 0, N2 = 0, C = 0, ndof = 0, nnz = 0;
  fsi::Scheme scheme{};
  int nfluid = 0, nsolid = 0;
  fsi::FluidProps fluid[fsi::MAX_REGIONS];
  fsi::SolidProps solid[fsi::MAX_REGIONS];

  // numbering: solver index <-> user index
  std::vector<int32_t> h_user2solver;        // [ndof]
  fsi::DevBuf<int32_t> user2solver, solver2user;

  // mesh / element data (device)
  fsi::DevBuf<double> geom;                  // [C][10]: Jinv (row-major dxi_k/dx_j) 9 + |det|
  fsi::DevBuf<int32_t> cell_dofs;            // [C][64] solver indices
  fsi::DevBuf<int32_t> cell_kind, cell_region;
  fsi::DevBuf<uint16_t> enbr;                // [C][10][10] index of node b in adj(node a)
  fsi::DevBuf<uint16_t> epnbr;               // [C][10][4]  index of vertex b among vertex-neighbours of node a
  fsi::DevBuf<int32_t> cell_rank;            // [C][10] rank of the local nodes
  fsi::DevBuf<int32_t> cell_prow;            // [C][4] pressure dofs (solver layout): with cell_rank, 56 bytes per cell give the 64 local dofs
  // Reproducible assembly (the default; FSI_ASSEMBLY=atomic switches both off).  Residual: element vectors stored to
  // Re[C][64] and summed per dof over the node's incident cells in ascending order (k_residual_gather).  Jacobian: the cells
  // are coloured (cells of one colour share no node) and every colour is one launch, so no matrix entry receives two adds
  // whose order could vary.
  fsi::DevBuf<double> Re;                    // [C][64]
  fsi::DevBuf<int64_t> inc_ptr, pinc_ptr;    // [N2 + 1], [V + 1]
  fsi::DevBuf<int32_t> inc, pinc;            // 16 * cell + local node
  int ncellcol = 0;                          // 0: single launch, unordered atomics (FSI_ASSEMBLY=atomic)
  fsi::DevBuf<int32_t> col_cells;            // [C] cell ids sorted by colour
  std::vector<int64_t> h_col_ptr;            // [ncellcol + 1]

  // node graph + CSR structure
  std::vector<int64_t> h_nadj_ptr;           // [N2+1]
  std::vector<int32_t> h_nadj;               // neighbour ranks, ascending
  std::vector<int64_t> h_padj_ptr;           // [N2+1] vertex-neighbours (as vertex ids), ascending
  std::vector<int32_t> h_padj;
  std::vector<int32_t> h_rank2node, h_node2rank;
  std::vector<int32_t> h_prank;               // vertex -> position in the pressure block
  std::vector<fsi::Level> levels;
  int ncolors = 0;
  bool coloured = false;                     // multicolour node ordering (needed by the ILU(0) kernels) or mesh order
  fsi::DevBuf<int64_t> nadj_ptr, padj_ptr;
  fsi::DevBuf<int32_t> nadj, padj;
  fsi::DevBuf<int64_t> rowptr;               // [ndof+1]
  fsi::DevBuf<int32_t> cols;                 // [nnz]
  fsi::DevBuf<int64_t> diagpos;              // [ndof]

  // matrices
  fsi::DevBuf<double> A_pre, A, LU;          // [nnz] each; A holds the row-equilibrated Jacobian after setup
  fsi::DevBuf<double> rowscale;              // [ndof]
  bool have_jacobian = false;

  // vectors (solver ordering)
  fsi::DevBuf<double> U, U1, F, b, du, bs, tmp1, tmp2, tmp3, tmp4, tmp5, tmp6, tmp7;
  fsi::DevBuf<double> scratch;               // reductions
  fsi::DevBuf<double> cellvals;              // per-cell diagnostics (fsi_flow_stats)
  fsi::DevBuf<int32_t> gv_idx;               // fsi_get_values: solver indices of the requested dofs
  fsi::DevBuf<int32_t> iflags;               // device-side error / counters

  // boundary data
  int64_t nbc = 0;
  fsi::DevBuf<int32_t> bc_dofs;              // solver indices
  fsi::DevBuf<double> bc_vals;
  int64_t npf = 0;                           // interface-pressure load: F[dof] += P * coef
  fsi::DevBuf<int32_t> pf_dofs;
  fsi::DevBuf<double> pf_coef;
  double P = 0.0;
  int64_t nrobin = 0;                        // Robin entries sorted by (row, col), duplicates merged: cols (d or v dofs), values
  int64_t nrobin_rows = 0;                   // distinct rows (v dofs) rb_urow, entries of row k: rb_ptr[k] .. rb_ptr[k+1]
  fsi::DevBuf<int32_t> rb_urow, rb_ptr;
  fsi::DevBuf<int32_t> rb_row, rb_col;
  fsi::DevBuf<double> rb_val;
  fsi::DevBuf<int64_t> rb_pos;               // position in the CSR values

  // element partition (fsi_set_partition): ghost rows, halo lists, the caller's transport
  bool part = false;
  int64_t C_owned = 0, nghost = 0, nsend = 0;
  std::vector<int32_t> h_bc, h_ident;        // solver indices
  int64_t nident = 0;                        // ghost dofs with incomplete rows (outermost layer): identity rows
  fsi::DevBuf<int32_t> ghost_idx, ident_idx, send_idx, mbc_dofs;   // mbc = Dirichlet + identity rows of the matrix
  int64_t nmbc = 0;
  fsi::DevBuf<double> ghost_zero;
  double *sendbuf = nullptr, *recvbuf = nullptr;
  FsiComm comm{};
  int64_t halo_calls = 0, allreduce_calls = 0;
  // library-side RCCL transport (fsi_set_rccl, fsi_rccl.hip): communicator, per-peer halo counts (doubles), staging buffer
  bool rccl = false;
  bool rccl_dead = false;                    // a RCCL call failed and the communicator was aborted: every later collective fails at once
  void* rccl_comm = nullptr;
  int rccl_rank = 0, rccl_world = 1;
  std::vector<int64_t> rccl_send, rccl_recv;
  fsi::DevBuf<double> rccl_red;
  int64_t rccl_allreduces = 0, rccl_halos = 0;
  bool ras = false;                          // restricted additive Schwarz on the overlap: agreed by all ranks in fsi_set_partition
  bool debug_gcr = false;
  int64_t dbg_droppable = 0, dbg_drop_cols = 0;
  int64_t dbg_cols = 0, dbg_sig6 = 0, dbg_sig9 = 0, dbg_sig12 = 0;   // FSI_DEBUG_GCR: how many Gram-Schmidt coefficients matter

  // field blocks for the block preconditioner (fsi_block.hip)
  int precond = 0;                           // 0 = field-split block preconditioner, 1 = monolithic multicolour ILU(0)
  bool have_monolithic_lu = false;
  bool prec_bad = false;                     // the preconditioner self-test failed on the current Jacobian
  fsi::DevBuf<int32_t> node_solid;           // [N2] by rank
  fsi::DevBuf<int32_t> vrank;                // [V] rank of the node of pressure position q
  fsi::DevBuf<int64_t> rowptr3, diagpos3;    // 3x3-blocked node structure (A_dd, Avv~, A_dv)
  fsi::DevBuf<int32_t> cols3;
  fsi::DevBuf<int64_t> rowptr_vp, rowptr_pv, rowptr_pp, diagpos_pp;
  fsi::DevBuf<int32_t> cols_vp, cols_pv, cols_pp;
  fsi::DevBuf<double> Adv, Avp, Apv, App;
  fsi::DevBuf<int64_t> s_rowptr, s_diagpos;  // explicit Schur complement on its full (two-ring) vertex pattern
  fsi::DevBuf<int32_t> s_cols;
  fsi::DevBuf<double> s_vals;
  fsi::DevBuf<float> s_vals32;                        // FSI_SCHUR_FP32: 1 (default) matrix values in FP16 / FP32, vectors FP64, product fused
  int schur_fp32 = 1;                                 // with the Chebyshev update; 0: FP64 values
  bool schur_tiled = false; int s_tile_max_nu = 0; int schur_tile = 256;   // FP16 records + tile-local columns for the Schur sweep (k_sweep_schur_tiled)
  fsi::DevBuf<uint32_t> s_rec; fsi::DevBuf<uint16_t> s_ploc; fsi::DevBuf<int64_t> s_tile_uptr; fsi::DevBuf<int32_t> s_tile_ulist;
  fsi::DevBuf<double> s_dinv;
  fsi::DevBuf<double> dd_db, vv_db;          // component-diagonal node-block copies of A_dd and Avv~ ([pairs][3])
  fsi::DevBuf<double> adv_db;
  bool dd_is_db = false, adv_is_db = false, adv_solid_only = false;      // (A_dv has entries in solid rows only: checked at every refresh)
  fsi::DevBuf<float> dd_db32, vv_db32, dd_dinv32, vvf_dinv32;   // FP32 copies for the Chebyshev sweeps
  int sweeps_fp32 = 1;
  fsi::DevBuf<float> dd_chat, ones32;         // scalar form of the Jacobi-scaled A_dd (one ratio per node pair)
  fsi::DevBuf<uint8_t> dd_rowflag;
  bool dd_is_scalar = false;
  int tile_max_nu = 0, tile_nodes = 256;
  bool tiled = false;                        // LDS-tiled sweep kernels usable (every tile's neighbour set fits the LDS tile)
  fsi::DevBuf<uint16_t> tile_ploc;           // [pairs] local index of the pair's column node in its tile's list
  fsi::DevBuf<int64_t> tile_uptr;            // [tiles+1]
  fsi::DevBuf<int32_t> tile_ulist;           // distinct neighbour nodes of each tile, ascending
  // two-level (P2 -> P1) solve of the displacement block: Galerkin coarse operator on the vertex graph (fsi_block.hip)
  int dd_mg = 1;                             // FSI_DD_MG=0: one-level Chebyshev sweeps
  bool mg_ready = false;
  int64_t mg_nc = 0, mg_cnnz = 0;            // coarse nodes = vertices in rank order
  fsi::DevBuf<int32_t> mg_par, mg_ccol, mg_child, mg_cfine;   // [N2][2] parents; coarse columns; children; vertex -> fine rank
  fsi::DevBuf<float> mg_pw, mg_chw;          // [N2][2] parent weights (vertex 1,0; edge node 1/2,1/2); child weights
  fsi::DevBuf<int64_t> mg_cptr, mg_chptr;
  fsi::DevBuf<double> mg_Ac;
  fsi::DevBuf<float> mg_cc, mg_d0, mg_dcinv4, mg_cones, mg_work;
  fsi::DevBuf<uint8_t> mg_cflag;
  int mg_pre = 3, mg_post = 5, mg_cits = 16;  // fine Chebyshev sweeps before / after the coarse solve; coarse sweeps (round 2: 4, 6, 40 on the Gershgorin interval)
  double mg_alpha = 20.0, mg_ckappa = 250.0, mg_clmax = 2.0;   // smoothing interval [lmax/alpha, lmax]; coarse interval
  fsi::SubMat Mdd, Mvv;                      // A_dd, Avv~
  fsi::DevBuf<double> blk;                   // work vectors of the block preconditioner
  int64_t nS = 0;                            // solid (incl. interface) nodes; compact velocity block A_SS on them
  fsi::DevBuf<int32_t> snode, ss_cols;
  fsi::DevBuf<int64_t> ss_rowptr, ss_diagpos, ss_src;
  fsi::DevBuf<double> ss_vals;
  int64_t sb_nblocks = 0;                    // FP32 block-CSR copy of A_SS used by the Chebyshev sweeps
  fsi::DevBuf<int64_t> sb_ptr, sb_src;
  fsi::DevBuf<int32_t> sb_col, sb_row, sb_stride;
  fsi::DevBuf<float> sb_vals, sb_dinv;
  int solid_fp32 = 1;
  int solid_block_jacobi = 1;                // 3x3 node-block scaling of the solid sweeps (FSI_SOLID_BJ=0: point Jacobi)
  // two-level solve of the solid velocity block (3x3-block version of the displacement cycle); FSI_SOLID_MG=0: 300 one-level sweeps
  int solid_mg = 1;
  bool sbmg_ready = false;
  int64_t sbmg_nc = 0, sbmg_nblk = 0;
  std::vector<int32_t> h_snode, h_sb_col;    // host copies for the hierarchy set-up
  std::vector<int64_t> h_sb_ptr;
  fsi::DevBuf<int32_t> sbmg_par, sbmg_ccol, sbmg_child, sbmg_cfine;
  fsi::DevBuf<float> sbmg_pw, sbmg_chw, sbmg_cvals, sbmg_cbinv12, sbmg_work;
  fsi::DevBuf<int64_t> sbmg_cptr, sbmg_chptr;
  fsi::DevBuf<uint8_t> sbmg_flag, sbmg_cflag;
  fsi::BcrData* bcr = nullptr;               // exact coarse solve (block cyclic reduction over BFS levels of the solid vertices), or null
  int solid_coarse_exact = 1;                // FSI_SOLID_COARSE_EXACT=0: the coarse level keeps its sbmg_cits Chebyshev sweeps
  int64_t bcr_solves = 0;
  const double* xs_zeroed = nullptr;         // the solid predictor's full-length vector whose non-solid entries are known to be zero (precondition_block)
  int sbmg_pre = 16, sbmg_post = 16, sbmg_cits = 90;      // round 2: 200 coarse sweeps on an interval that ended at 2.2x the largest eigenvalue
  double sbmg_alpha = 200.0, sbmg_ckappa = 4000.0, sbmg_clmax = 2.0;
  int64_t nfs = 0;                           // fluid-interior velocity rows with solid columns (coupling of the predictor)
  fsi::DevBuf<int32_t> fs_rows, fs_col;
  fsi::DevBuf<int64_t> fs_ptr, fs_src;
  int solid_fused = 1;                       // SpMV + Chebyshev update of a solid sweep in one launch (FSI_SOLID_FUSED=0: two)
  fsi::DevBuf<float> sb_binv12;
  fsi::DevBuf<double> sb_binv9;
  fsi::DevBuf<double> mask_s, mask_f;        // [3 N2] 1 on velocity dofs of solid (incl. interface) / fluid-interior nodes
  int cheb_its_s = 300, cheb_its_f = 4, cheb_its_p = 30, cheb_its_d = 60;     // Chebyshev sweeps on the solid / fluid part of the velocity block
  double cheb_kappa_s = 1e4, cheb_kappa_f = 5.0, cheb_kappa_p = 100.0, lmax_s = 1.0, lmax_f = 1.0, lmax_p = 1.0, cheb_kappa_d = 1000.0, lmax_d = 1.0;
  int64_t inner_its[3] = {0, 0, 0};          // accumulated inner iterations: vv, schur, dd
  int64_t inner_calls = 0;
  int64_t pivot_warnings = 0;

  // Krylov recycling space (GCR): Q = A P (orthonormal; FP32 or FP64 storage) and the directions P.  Directions made
  // during the current solve are held as raw preconditioned vectors in KZ plus coefficient columns on the host
  // (kry_cn), and become explicit at the next flush (fsi_gcr.hip).
  int64_t kry_cap = 0, kry_m = 0;            // kry_m: directions made since the last Jacobian (statistics / ring age)
  int64_t kry_hw = 0;                        // slots in use (columns scanned by the kernels), <= kry_cap
  int kry_fp32 = 0;                          // storage of Q during the current Jacobian lifetime: 0 FP64, 1 FP32
  int kry_fp32_policy = 2;                   // (3: was 2, fell back to FP64 after a failed cycle)  FSI_KRYLOV_FP32: 0 never, 1 always, 2 (default) decided by the first solve after a
                                             // Jacobian refresh: FP32 iff that solve's tolerance is >= 1e-7 (see solve_gcr)
  int64_t ldq = 0, ldz = 0;                  // column strides (elements)
  fsi::DevBuf<unsigned char> KQ;             // [kry_cap][ldq] float or double
  fsi::DevBuf<double> KZ;                    // [kry_cap][ldz]
  // FP32 storage only: the directions of the current solve cycle also in FP64 (a window of 32).  A M^-1 r_k is nearly
  // parallel to the previous direction after a step of little progress - a cancellation of 1e5 and more, which is
  // taken out exactly against this window before the FP32 columns see the vector.
  fsi::DevBuf<double> KQh, hcoef_hot;        // [32][ldq], [40]
  std::vector<int32_t> hot_slots;            // slot of each window column (-1: empty)
  int hot_next = 0;
  fsi::DevBuf<double> hcoef;                 // [kry_cap + 2] device: h = Q^T w, w.w, w.r
  fsi::DevBuf<double> gcr_out;               // [8] device: small reduction results
  fsi::DevBuf<double> gcr_y, gcr_cn;         // [kry_cap], [32][kry_cap] device copies of the flush coefficients
  fsi::DevBuf<int32_t> gcr_slots;            // [32]
  double* gcr_host = nullptr;                // pinned [kry_cap + 16]
  std::vector<int64_t> kry_born;             // [kry_cap] creation index of the direction in each slot (-1: free)
  std::vector<int32_t> kry_free;             // free slots (retired in batches when the store is full)
  double gs_rtol = 0.0;                      // tightest linear tolerance asked for since the last Jacobian (re-orthogonalisation criterion)
  int64_t ortho_q_cols = 0, ortho_z_cols = 0, ortho_q_launches = 0, ortho_z_launches = 0;   // columns streamed (exact bytes of the orthogonalisation)
  fsi::DevBuf<float> A32;                         // FP32 copy of A for the products inside the Krylov iterations (FSI_OPERATOR_FP32, default on)
  bool op32_ok = false; int op32_policy = 1; int64_t op32_products = 0;
  fsi::DevBuf<int64_t> a32_ptr;              // [N2 + 1] first entry of a node's padded block in A32 (k_spmv_node6p)
  fsi::DevBuf<int32_t> a32_cols;             // padded index rows
  int64_t a32_ptail = 0, a32_tail_src = 0, a32_tail_nnz = 0;     // pressure rows: behind the padded node blocks, unpadded
  // d rows of the node blocks in pair form (k_drows_extract: six values per node pair instead of 18; FsiTuning.compact_drows):
  // refreshed and CHECKED with every Jacobian - drows_ok only while nothing else was found in those rows
  fsi::DevBuf<double> Ad64;                  // [6 x node pairs]
  fsi::DevBuf<float> Ad32;                   // ... rounded, for the products on the FP32 copy
  bool drows_ok = false;
  int64_t drows_products = 0;
  bool gcr_stagnated = false;                // the last cycle ended on 40 iterations without a 10 % gain
  bool gcr_stalled = false;                  // ... on 80 iterations without a 10 % gain far from its target (truncated recurrence stuck)
  double f32_cycle_floor = 1e-6;             // FP32 basis: a cycle reduces the residual by at most this factor before the FP64 verdict
  double f32_verdict_skip_rtol = 3e-4;       // FP32 basis: answers asked for at or above this may skip the FP64 verdict (see solve_gcr)
  double f32_last_drift = -1.0;              // |true - recurrence residual| / |b| of the last verified FP32 cycle on the present store
  int64_t verdicts_skipped = 0;
  bool in_newton = false;                    // the running fsi_solve was called by fsi_newton_solve (an FP64 Newton residual follows)
  bool f64_suspect = false;                  // FP64 basis: a verdict of the present store differed from its recurrence (see solve_gcr)
  int64_t ev_base[3] = {0, 0, 0};            // newton_retries / fp32 fall-backs / gcr_restarts counted before the last timer reset (fsi_get_solver_events)
  int64_t newton_retries = 0;                // Newton iterations whose linear solve failed on a stale Jacobian and succeeded after a refresh
  double orth_floor32 = 3e-7, orth_floor64 = 1e-9;   // estimated orthogonality error of a new column above which a second Gram-Schmidt pass is made
  int64_t gcr_reorth_forced = 0;
  int cheb4 = 1;                             // bit 0 / 1: 4th-kind Chebyshev smoothing sweeps in the solid / displacement two-level cycle
                                             // (scan in DESIGN.md section 5: solid -1.4 % of a step at the same 16 + 16 sweeps, displacement worse)
  double gcr_escape = 1e-3;                  // alpha^2 <= this * |r|^2: the direction did not move the residual, next one from q
  int64_t kry_fp32_failures_total = 0;       // fall-backs from the FP32 basis since the timers were reset
  int64_t gcr_arnoldi_steps = 0;             // directions made from the last q because the residual had not moved (see gcr_cycle)
  int64_t gcr_restarts = 0;                  // solves that dropped the kept directions and restarted because of that
  int debug_prec_apply = 0;
  fsi::DevBuf<float> Avp32, Apv32; bool pv32_ok = false;     // FSI_PV_FP32 (default on): FP32 copies for k_vel_correct32 / k_pres_rhs32
  fsi::DevBuf<double> vv_dinv;               // [3 N2] 1 / diag(Avv~), for the velocity correction
  fsi::DevBuf<uint8_t> adv_rowmask;          // [N2] 1: the node's A_dv rows have entries (solid nodes), 0: all zero, not streamed
  bool sweeps_fp16 = true;                   // FSI_SWEEPS_FP16=0: FP32 matrix values in the fine-level sweeps (k_sweep_tiled_f32 / k_sweep_sb_b3)
  fsi::DevBuf<uint32_t> dd_rec, vv_rec, sb_rec;   // packed FP16 records: [pairs], [pairs][2], [blocks][6] 32-bit words
  bool fused_sweeps = true;                  // FSI_FUSED_SWEEPS=0: product and Chebyshev update of the FP32 sweeps as two launches

  float sbmg_gersh = 2.f, mg_gersh = 2.f;     // the row-sum bounds of the two coarse levels (fallback of the self-test)
  int coarse_power = 1;                      // FSI_COARSE_POWER=0: coarse levels' Chebyshev intervals end at the Gershgorin bound (round 2)
  bool dd_same = false, dd_checksum_valid = false;   // this refresh found the displacement block unchanged / a checksum exists
  int64_t dd_cache_hits = 0;                 // refreshes that kept the displacement block's coarse operator and eigenvalue estimate
  double lmax_d_cached = 0.0, dd_checksum[3] = {0.0, 0.0, 0.0};   // largest eigenvalue of the (constant) displacement block, and what it was computed for
  int64_t part_allreduces = 0;               // partitioned runs: all-reduces issued inside the Krylov iterations (tests count them)
  int kry_fp32_failures = 0;                 // cycles that lost the system in FP32 storage (policy 2 -> 3); two of them pin FP64
  double tol_hint = 0.0, bnorm_max = 0.0;     // fsi_newton_solve -> solve_gcr: lowest linear tolerance to expect; largest |b| seen
  double gcr_reorth = 0.0;                   // FSI_GCR_REORTH: second Gram-Schmidt pass when |w'| < reorth |w| (0: automatic)
  double newton_forcing_late = 3e-3;         // forcing term of late Newton iterations (see fsi_newton_solve); >= newton_forcing or 0: off.
                                             // Round 4 scan (profiles/r04_forcing_scan.txt): distance of the known-answer run to the exact-solve
                                             // trajectory 4.7e-6 -> 1.4e-6 in v for +3 % of the bench's time step (2e-3: 1.0e-6 / +7 %; 1e-3: 5.6e-7 / +8 %)
  double newton_adaptive = 0.3;              // forcing term from the contraction the Newton iteration of the same index reached one time step ago (see fsi_newton_solve); 0: off
  double nw_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // that contraction, net of the linear residual that was allowed; 0 = not known (reset with the Jacobian)
  int64_t newton_adaptive_solves = 0;
  // adaptive solves only (set by fsi_newton_solve around its fsi_solve, zero otherwise): the answer must also leave an UNSCALED
  // residual below utol * b_unscaled - Newton's policy reads the unscaled |b|, the Krylov method minimises the row-equilibrated
  // one, and the rows differ by twelve decades - or the tolerance is tightened towards utol_rtol_floor (the non-adaptive value)
  double utol = 0.0, utol_rtol_floor = 0.0, b_unscaled = 0.0;
  double utol_ratio = 0.0;                   // (unscaled / scaled) relative residual at the last check under this Jacobian: where the next adaptive solve starts
  int64_t utol_tightened = 0;
  double newton_late_factor = 10.0;          // "late": the previous update norm (or |b|) is within this factor of its tolerance
  int64_t newton_late_solves = 0;
  // Two chains of one preconditioner application side by side (FSI_PREC_STREAMS=1; precondition_block): stream A = solver
  // stream: split, solid predictor, displacement block, merge; stream B: fluid predictor, pressure step, velocity correction.
  int prec_streams = 1;                      // FSI_PREC_STREAMS=0: one chain on the solver stream (rounds 1-3)
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_split = nullptr, ev_solid = nullptr, ev_b = nullptr;
  int dd_early = 0;                          // FSI_DD_EARLY=1: displacement rhs from the solid predictor (measurement)
  int vel_jacobi = 0;                        // FSI_VEL_JACOBI=1: no solid -> fluid coupling inside the velocity predictor (measurement)
  double newton_forcing = 1e-2;              // inexact Newton: linear tolerance = forcing * atol / |b| (FSI_NEWTON_FORCING; 1e-2: same Newton counts as 1e-3 on the bench, 18 % fewer Krylov iterations)

  // timers
  fsi::PhaseTimer t_res, t_jac, t_fac, t_spmv, t_prec, t_ortho, t_flush, t_kry, t_ss;
  fsi::PhaseTimer t_sch;                     // sampled Schur-complement sweeps
  hipEvent_t sch_ev0[4] = {}, sch_ev1[4] = {};
  int sch_samples_pending = 0;
  hipEvent_t ss_ev0[8] = {}, ss_ev1[8] = {};
  int ss_samples_pending = 0;
  fsi::PhaseTimer t_db;
  hipEvent_t db_ev0[8] = {}, db_ev1[8] = {};
  int db_samples_pending = 0;
  fsi::PhaseTimer t_sc;
  hipEvent_t sc_ev0[4] = {}, sc_ev1[4] = {};
  int sc_samples_pending = 0;
  int64_t kry_iters = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int sample_budget = 0;                     // preconditioner applications whose sweep kernels are still sampled with events

  // host copy of the mesh needed after create
  std::vector<double> h_coords;
  std::vector<int32_t> h_tet_nodes;
};
