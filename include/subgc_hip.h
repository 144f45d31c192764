/*
 * subgc_hip.h -- C ABI of libsubgc_hip.so: the MI355X (gfx950) kernels of the Sub-GC hot path.
 *
 * The reference (YiwuZhong/Sub-GC) has no FFI layer: its hot path is ATen calls made from
 * models/AttModel.py, models/lib/{gcn_backbone,graph_conv,graph_conv_unit,gpn}.py and
 * misc/utils.py.  Each entry point below replaces the op site cited next to it (file:line in
 * /root/reference).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Contract for EVERY function:
 *   - returns 0 (SUBGC_OK) or a negative SUBGC_E* code; never throws, never allocates device
 *     memory, never synchronises the device; all work is enqueued on `stream` (a hipStream_t
 *     passed as void*; NULL = the legacy default stream);
 *   - all pointers are BORROWED device pointers owned by the caller (e.g. tensor.data_ptr());
 *     outputs / scratch are caller-allocated; row-major, fp32 unless noted, indices int64 where
 *     the reference uses LongTensor inputs and int32 for internal index structures;
 *   - re-entrant per stream: there is NO process-global scratch, mode or tuning state (no environment variable is read; the ONE
 *     exception is the opt-in profiling hook subgc_prof_enable / subgc_prof_collect below, a process-wide event log for bench.py) -- the entry points that can use a split-K scratch
 *     take `workspace, ws_bytes` as CALL arguments (subgc_gemm_workspace_bytes / subgc_gemm_bf16_workspace_bytes say how much
 *     a shape can use; NULL / smaller is legal), so calls on different streams only need different workspaces;
 *     subgc_last_error() returns a thread-local message for the last failing call on the calling thread;
 *   - "bf16" tensors are raw uint16 bit patterns (torch.bfloat16 storage); an `*_bf16` int argument says that the named
 *     destination (or source) pointer, declared void*, holds bf16 instead of fp32.
 */
#ifndef SUBGC_HIP_H
#define SUBGC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUBGC_OK 0
#define SUBGC_EINVAL (-1)   /* bad argument (null pointer, negative size, unsupported shape) */
#define SUBGC_ELAUNCH (-2)  /* hipLaunch / runtime error, see subgc_last_error() */
#define SUBGC_EALIGN (-3)   /* pointer / leading dimension violates a documented alignment */

#define SUBGC_ABI_VERSION 1
int subgc_version(void);
const char* subgc_last_error(void);
/* name of the gfx target the kernels were compiled for ("gfx950") */
const char* subgc_arch(void);
/* Debug bounds mode (off by default; returns the previous setting).  While on, every entry point that consumes an index tensor it did not
 * produce itself -- rel_ind (csr_build, gcn_*), gpn_obj_ind (gpn_prep / gpn_test_prep / gpn_select, with the reference's node-list-vs-mask
 * consistency assert, gpn.py:117-118), sub-graph node lists (subgraph_pool_*, subgraph_nms*, pack_rows), token ids (embed_*, token_rows),
 * criterion targets (masked_nll_*, nll_logsoftmax_bwd), class ids (class_partials) -- validates it on the device BEFORE launching its
 * kernels and returns SUBGC_EINVAL with the tensor's name, the first offending position and value in subgc_last_error(), as the
 * reference fails with an IndexError / AssertionError on a bad loader tensor.  Costs one launch and one stream synchronisation per
 * checked tensor; skipped on a capturing stream.  With the mode off the kernels keep their documented clamping behaviour.           */
int subgc_debug_bounds(int on);

/* ---- profiling hook used by bench.py (HIP events on the launch stream) -------------------
 * While enabled, every launch of kernel family `family` (see SUBGC_FAM_*) is bracketed by a
 * pair of hipEvents recorded on its stream.  subgc_prof_collect() synchronises those events,
 * returns the number of launches and their summed duration (ms) and summed `work` units the
 * launches reported (flops for GEMM, bytes for the HBM-bound families), then clears the log. */
#define SUBGC_FAM_GEMM 1
#define SUBGC_FAM_ATTN 2
#define SUBGC_FAM_LSTM 3
#define SUBGC_FAM_GCN 4
#define SUBGC_FAM_POOL 5
#define SUBGC_FAM_SOFTMAX 6
int subgc_prof_enable(int family, int on);
int subgc_prof_collect(int family, int64_t* launches, double* total_ms, double* total_work);
/* bytes the family's launches of the LAST subgc_prof_collect actually moved, where an entry point reports them separately from its
 * algorithmic `work` (the LSTM cell kernels: split-K planes of the gate products, additive gate terms and saved gates included) */
int subgc_prof_last_moved(int family, double* moved_bytes);

/* ======================================================================================
 * Dense contractions (MFMA v_mfma_f32_32x32x2_f32; exact fp32)
 * replaces: every nn.Linear / LSTMCell matmul on the path -- AttModel.py:363-366,376-377,386,
 *           411-413,421-423,336,340,453; graph_conv_unit.py:29-30; gpn.py:54,79 -- and their
 *           autograd backward.
 *
 *   C[M,N] = epilogue( op(A)[M,K] * op(B)[K,N] )
 *   transA = 0: A stored [M,K] (lda >= K)      transA = 1: A stored [K,M] (lda >= M)
 *   transB = 0: B stored [K,N] (ldb >= N)      transB = 1: B stored [N,K] (ldb >= K)  (nn.Linear weight)
 *   epilogue, in this order:  v = acc
 *        + bias[n]                       (bias  != NULL)
 *        + add[row, n]                   (add   != NULL, leading dim ldadd)
 *        v = max(v, 0)                   (flags & SUBGC_GEMM_RELU)
 *        v *= keep[row, n] * keep_scale  (keep  != NULL: uint8 dropout keep-mask, ld = ldc)
 *        v += C[row, n]                  (flags & SUBGC_GEMM_ACCUM)
 *        C[row, n] = v
 *   a_rows (int32[M], transA = 0 only): row m of op(A) is A[a_rows[m], :]; a negative index
 *        reads a zero row.   c_rows (int32[M]): `row` above is c_rows[m] instead of m (rows with
 *        a negative c_rows[m] are not written).   m_dev (int32*, device): the number of valid
 *        ROWS OF THE STORED A matrix, read on the device so ragged row sets need no host round
 *        trip: with transA = 0 the effective M is min(M, *m_dev); with transA = 1 (A stored
 *        [K,M], the weight-gradient form dW = dY^T X over a ragged row set) the effective K is
 *        min(K, *m_dev).
 */
#define SUBGC_GEMM_RELU 1
#define SUBGC_GEMM_ACCUM 2
/* arithmetic of the 128x128-tile forms, chosen PER CALL in bits 4-5 of `flags` (0 = the process default, which is fp32
 * unless the environment says SUBGC_GEMM_X3=1|2 at first use):
 *   F32     v_mfma_f32_32x32x2_f32 on fp32 operands;
 *   BF16X3  each fp32 operand is split EXACTLY into three bf16 planes and the product is formed from six
 *           v_mfma_f32_32x32x16_bf16 terms accumulated in fp32 (csrc/gemm_x3.h): fp32-level accuracy at 2.67x the pipe rate;
 *   BF16R   each fp32 operand is rounded to nearest-even bf16 on its way to LDS, one bf16 MFMA term (storage stays fp32;
 *           the bf16-STORAGE path of BASELINE configs 3 and 5 is subgc_gemm_bf16 below).                                   */
/* measurement-script switches, per call like everything else (the library reads no environment variable and keeps no tunable
 * state): keep subgc_gemm_f32 off its split-K forms / off the weight-streaming form for M <= 80; force subgc_gemm_bf16's tile */
#define SUBGC_GEMM_NO_SPLITK (1 << 6)
#define SUBGC_GEMM_NO_SKINNY (1 << 7)
#define SUBGC_GEMM_TILE128 (1 << 6)
#define SUBGC_GEMM_TILE256 (1 << 7)
#define SUBGC_GEMM_SPLITS(n) (((n) & 15) << 8)    /* subgc_gemm_bf16: force n K parts (0 = the cost model decides) */
#define SUBGC_GEMM_SPLITS_OF(flags) (((flags) >> 8) & 15)
#define SUBGC_GEMM_NO_ROW_CUT (1 << 12)            /* subgc_gemm_bf16: one launch even for near-whole-round tile counts */
#define SUBGC_GEMM_TILE_P8 (1 << 13)               /* subgc_gemm_bf16: force the 256 x 256 x 64 eight-phase form (measurement scripts) */
#define SUBGC_GEMM_NO_P8 (1 << 14)                 /* subgc_gemm_bf16: never choose the eight-phase form (A/B timing) */
#define SUBGC_GEMM_MODE_F32 (1 << 4)
#define SUBGC_GEMM_MODE_BF16X3 (2 << 4)
#define SUBGC_GEMM_MODE_BF16R (3 << 4)
/* workspace / ws_bytes: caller-owned scratch for the split-K forms of THIS call (shapes whose 128x128 tile count cannot
 * fill the 256 CUs, e.g. the per-step recurrent GEMMs with M = 640, write partial tiles there and reduce them, all ordered
 * on `stream`).  NULL / 0: those shapes run with 64x64 tiles.  Concurrent calls on different streams need different
 * workspaces; calls on one stream may share one.                                                                      */
int subgc_gemm_f32(int transA, int transB, int M, int N, int K,
                   const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                   const float* bias, const float* add, int64_t ldadd,
                   const uint8_t* keep, float keep_scale, int flags,
                   const int32_t* a_rows, const int32_t* c_rows, const int32_t* m_dev,
                   void* workspace, size_t ws_bytes, void* stream);
/* Two subgc_gemm_f32 products of the SAME shape, layout, leading dimensions and epilogue in ONE launch: (A1, B1 -> C1) in the first half of
 * the grid, (A2, B2 -> C2) in the second (the two collection units of a GCN pair, models/lib/graph_conv.py:24-25,31-32: their d(H) halves and
 * their fc_rgt weight gradients are half-filling launches one by one).  Epilogue: bias1 / bias2, SUBGC_GEMM_ACCUM; SUBGC_GEMM_MODE_* as
 * subgc_gemm_f32.                                                                                                                       */
int subgc_gemm_f32_pair(int transA, int transB, int M, int N, int K, const float* A1, const float* A2, int64_t lda, const float* B1,
                        const float* B2, int64_t ldb, float* C1, float* C2, int64_t ldc, const float* bias1, const float* bias2, int flags,
                        void* workspace, size_t ws_bytes, void* stream);
/* the most scratch subgc_gemm_f32 can use for a shape (0: it never splits) */
int subgc_gemm_workspace_bytes(int M, int N, int K, size_t* bytes);

/* column sums: out[n] (+)= sum_m X[m, n]  -- bias gradients.  accumulate != 0 adds to out.
 * workspace / ws_bytes (optional scratch of THIS call, 16-byte aligned, up to 1024 x N floats): tall matrices are then cut into
 * ~1024 short slabs whose partial sums are added in a fixed order by a second launch (no float atomics, reproducible); without
 * it 256-row slabs are merged with atomics.                                                                              */
int subgc_colsum_f32(const float* X, int64_t ldx, int M, int N, float* out, int accumulate,
                     const int32_t* m_dev, void* workspace, size_t ws_bytes, void* stream);
/* column sums of up to three bf16 matrices of ONE shape and leading dimension in two launches instead of six (a GCN unit pair's three bias
 * gradients); X1 / X2 and their destinations are ignored beyond n                                                                    */
int subgc_colsum_bf16_set(int n, const uint16_t* X0, const uint16_t* X1, const uint16_t* X2, int64_t ldx, int M, int N, float* out0,
                          float* out1, float* out2, int accumulate, void* workspace, size_t ws_bytes, void* stream);
/* Weight gradient AND bias gradient of one linear layer in one call (the backward of every nn.Linear / nn.LSTMCell product of
 * models/AttModel.py:363-366,376-377,386,411-413,421-423,453 and models/lib/graph_conv_unit.py:29-30; autograd's
 * `grad_weight = grad_output.t().mm(input)` and `grad_bias = grad_output.sum(0)`):
 *   dW[M, N] (+)= dY^T X   and   db[M] (+)= sum_k dY[k, :]      with dY stored [K, M], X stored [K, N] (rows = samples).
 * The workgroups that own tile column 0 of dW add up the dY tiles they stage anyway, so dY is not read a second time and the separate
 * column-sum launches (subgc_colsum_f32 + its finish pass) disappear; sums are formed in a fixed order (no float atomics).
 * flags: SUBGC_GEMM_ACCUM (dW is added to) and the SUBGC_GEMM_MODE_* bits (the opt-in arithmetics run as the two separate passes);
 * db_accumulate: db is added to; m_dev: device-side bound on K (live rows); workspace as subgc_gemm_f32 (its last 32 M bytes hold
 * the split-K forms' partial column sums).                                                                                        */
int subgc_gemm_f32_wgrad(int M, int N, int K, const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw,
                         float* db, int flags, int db_accumulate, const int32_t* m_dev, void* workspace, size_t ws_bytes, void* stream);
/* the same over bf16 rows: bias gradients from bf16-stored gate / logit gradients */
int subgc_colsum_bf16(const uint16_t* X, int64_t ldx, int M, int N, float* out, int accumulate,
                      const int32_t* m_dev, void* workspace, size_t ws_bytes, void* stream);

/* ======================================================================================
 * Index kernels (bit-exact)
 * row_argmax: out[r] = add + argmax_{c >= skip} X[r, c] - 0 ... first maximum wins.
 * replaces AttModel.py:376,383,385 (class argmax, `skip`=1,`add`=0 keeps the +1 of the
 * reference because indices are reported in the un-skipped frame), gpn.py:66 (slot select),
 * AttModel.py:306 (greedy token; also returns the max value when val != NULL).          */
int subgc_row_argmax_f32(const float* X, int64_t ldx, int rows, int cols, int skip,
                         int64_t* idx, float* val, void* stream);
/* the same arg-max with an int32 index output (the class ids that index the embedding tables, AttModel.py:374-385) */
int subgc_row_argmax_i32(const float* X, int64_t ldx, int rows, int cols, int skip, int32_t* idx, void* stream);

/* CSR of the scene graph by subject and by object (replaces gcn_backbone.py:55-67 make_map).
 * rel_ind int64 [B,K,2]; for role r in {0,1}: ptr[r][b, 0..N] (int32, [2,B,N+1]) and
 * edges[r][b, 0..K-1] (int32, [2,B,K]) with the relations of node n at
 * edges[r][b, ptr[r][b,n] .. ptr[r][b,n+1]) in ascending k.  Returns SUBGC_EINVAL on the
 * host for bad sizes; out-of-range node ids are clamped to N-1 on the device.           */
int subgc_csr_build(const int64_t* rel_ind, int B, int K, int N, int32_t* ptr, int32_t* edges, void* stream);

/* ======================================================================================
 * GCN aggregation (replaces graph_conv_unit.py:34-36 bmm + normalise + ReLU, graph_conv.py:26,33
 * averaging and gcn_backbone.py:43-47 residual).
 *
 * nodes <- relations (units 0,1):
 *   X'[b,n,:] = 1/2 relu( sum_{k: s_k=n} F0[b,k,:] / (cnt_s[n] + 1e-7) )
 *             + 1/2 relu( sum_{k: o_k=n} F1[b,k,:] / (cnt_o[n] + 1e-7) )  (+ skip[b,n,:])
 *   act (uint8 [B,N,L], bit0: subject-role pre-activation > 0, bit1: object-role) is saved for
 *   the backward:  dF0[b,k,:] = 1/2 [act&1](b,s_k,:) dX'[b,s_k,:] / (cnt_s[s_k] + 1e-7), same for F1.
 * relations <- nodes (units 2,3), c = fp32(1 + 1e-7):
 *   P'[b,k,:] = 1/2 relu(F2[b,s_k,:] / c) + 1/2 relu(F3[b,o_k,:] / c)  (+ skip[b,k,:])
 *   dF2[b,n,:] = 1/2 [F2[b,n,:] > 0] / c * sum_{k: s_k=n} dP'[b,k,:], same for F3 with o_k.
 */
int subgc_gcn_nodes_fwd(const float* F0, const float* F1, const int32_t* ptr, const int32_t* edges,
                        const float* skip, float* Xout, uint8_t* act, int B, int N, int K, int L, void* stream);
/* out_bf16 (nodes_bwd, edges_bwd_bn): the gradients are written bf16 -- the storage type of the unit outputs they belong to under
 * compute_dtype = bf16 (their only consumers are GEMMs) -- instead of fp32 followed by a cast pass. */
int subgc_gcn_nodes_bwd(const float* dX, const uint8_t* act, const int64_t* rel_ind, const int32_t* ptr,
                        void* dF0, void* dF1, int out_bf16, int B, int N, int K, int L, void* stream);
int subgc_gcn_edges_fwd(const float* F2, const float* F3, const int64_t* rel_ind, const float* skip,
                        float* Pout, int B, int N, int K, int L, void* stream);
int subgc_gcn_edges_bwd(const float* dP, const float* F2, const float* F3, const int32_t* ptr,
                        const int32_t* edges, float* dF2, float* dF3, int B, int N, int K, int L, void* stream);

/* BatchNorm1d over the rows of X[M,C] (graph_conv_unit.py:31-32; Full-GC only).
 * training != 0: batch statistics (biased variance for the normalisation, unbiased for the
 * running_var update, momentum 0.1, eps 1e-5), saves mean / rstd [C] for the backward.
 * workspace / ws_bytes (optional scratch of THIS call, 16-byte aligned; ~1024 x C floats, twice that for the backward): the
 * column reductions then leave per-slab partial sums there and add them in a fixed order (no atomics, reproducible
 * statistics); with a smaller or no workspace they fall back to float atomics.                                            */
int subgc_bn_fwd(const float* X, float* Y, int M, int C, const float* gamma, const float* beta,
                 float* running_mean, float* running_var, float* save_mean, float* save_rstd,
                 int training, float momentum, float eps, void* workspace, size_t ws_bytes, void* stream);
int subgc_bn_bwd(const float* dY, const float* X, const float* gamma, const float* save_mean,
                 const float* save_rstd, float* dX, float* dgamma, float* dbeta, int M, int C,
                 void* workspace, size_t ws_bytes, void* stream);

/* The same BatchNorm FUSED into the aggregation kernels that consume it (csrc/gcn_bn.hip; graph_conv_unit.py:31-36): the normalised
 * tensor is never written.
 *   subgc_bn_stats: one pass over the raw unit output X [M, C] (fp32, or bf16 when x_bf16 != 0; C % 4 == 0) -> aff [3, C] =
 *     {mean, gamma * rstd, beta} for the consumer and rstd [C] for the backward; running statistics updated as nn.BatchNorm1d
 *     does (momentum, unbiased variance).  training == 0: aff from the running statistics (X is not read, rstd may be NULL).
 *     workspace: subgc_bn_stats_workspace_bytes(M, C) bytes, 16-byte aligned (per-slab shifted sums, merged in double).
 *   subgc_gcn_{nodes,edges}_fwd_bn: subgc_gcn_{nodes,edges}_fwd with the sources read as (F - mean) * scale + beta through
 *     aff0/aff1 ([3, L] each; NULL = the source is used as it is), sources fp32 or bf16 (f_bf16), and an optional bf16 copy of
 *     the result (Xout16 / Pout16: the next layer's GEMM operand).  subgc_gcn_edges_bwd_bn: the ReLU sign test on the same
 *     normalised values.  L % 4 == 0.
 *   subgc_bn_bwd_fused: dY = d(normalised) [M, C] fp32 -> dX in X's storage type (dx_bf16) and dgamma / dbeta [C] (accumulate != 0:
 *     added to what is there); mean = aff[0:C].  workspace: 2/3 of the forward's.                                              */
/*   subgc_bn_stats_pair / subgc_bn_bwd_fused_pair: the TWO units a fused aggregation consumes (same M, C, storage type) in two launches
 *     instead of four each (training mode; workspace: twice the single call's).                                                    */
int subgc_bn_stats_pair(const void* X0, const void* X1, int x_bf16, int M, int C, const float* gamma0, const float* gamma1, const float* beta0,
                        const float* beta1, float* rmean0, float* rmean1, float* rvar0, float* rvar1, float* aff0, float* aff1, float* rstd0,
                        float* rstd1, float momentum, float eps, void* workspace, size_t ws_bytes, void* stream);
int subgc_bn_bwd_fused_pair(const float* dY0, const float* dY1, const void* X0, const void* X1, int x_bf16, int M, int C, const float* gamma0,
                            const float* gamma1, const float* mean0, const float* mean1, const float* rstd0, const float* rstd1, void* dX0,
                            void* dX1, int dx_bf16, float* dgamma0, float* dgamma1, float* dbeta0, float* dbeta1, int accumulate,
                            void* workspace, size_t ws_bytes, void* stream);
int subgc_bn_stats_workspace_bytes(int M, int C, size_t* bytes);
int subgc_bn_stats(const void* X, int x_bf16, int M, int C, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float* aff, float* rstd, int training, float momentum, float eps, void* workspace,
                   size_t ws_bytes, void* stream);
int subgc_bn_bwd_fused(const float* dY, const void* X, int x_bf16, int M, int C, const float* gamma, const float* mean,
                       const float* rstd, void* dX, int dx_bf16, float* dgamma, float* dbeta, int accumulate, void* workspace,
                       size_t ws_bytes, void* stream);
int subgc_gcn_nodes_fwd_bn(const void* F0, const void* F1, int f_bf16, const float* aff0, const float* aff1, const int32_t* ptr,
                           const int32_t* edges, const float* skip, float* Xout, uint16_t* Xout16, uint8_t* act, int B, int N, int K,
                           int L, void* stream);
int subgc_gcn_edges_fwd_bn(const void* F2, const void* F3, int f_bf16, const float* aff2, const float* aff3, const int64_t* rel_ind,
                           const float* skip, float* Pout, uint16_t* Pout16, int B, int N, int K, int L, void* stream);
int subgc_gcn_edges_bwd_bn(const float* dP, const void* F2, const void* F3, int f_bf16, const float* aff2, const float* aff3,
                           const int32_t* ptr, const int32_t* edges, void* dF2, void* dF3, int out_bf16, int B, int N, int K, int L,
                           void* stream);

/* ======================================================================================
 * sGPN (replaces gpn.py:152-185 gather + diagonal bmm + max/mean pooling, never materialising
 * the gathered [G,N,L] tensor).  For sub-graph g (image img[g]) with node list idx[g,0..N) and
 * per-slot weights w[g,i] (the diagonal of gpn_pool_mtx) and denom[g] (sum of att_masks):
 *   out[g, 0:L]  = max_i  w[g,i] * X[img[g], idx[g,i], :]      (all N slots, zeros included)
 *   out[g, L:2L] = sum_i  w[g,i] * X[img[g], idx[g,i], :] / denom[g]
 * argmax (int32 [G,L]) records the first maximising slot for the backward, which scatter-adds
 * into dX with fp32 atomics (node rows are shared between sub-graphs).
 * idx is read with element stride 1 and row stride idx_stride (int64 elements); w is read as
 * w[g*w_gstride + i*w_istride] so the diagonal of the caller's [.,N,N] pool matrix can be used
 * in place (w_gstride = N*N, w_istride = N+1).                                              */
int subgc_subgraph_pool_fwd(const float* X, const int64_t* idx, int64_t idx_stride, const float* w,
                            int64_t w_gstride, int64_t w_istride, const float* denom, const int32_t* img,
                            float* out, int32_t* argmax, int G, int N, int L, void* stream);
int subgc_subgraph_pool_bwd(const float* dout, const int64_t* idx, int64_t idx_stride, const float* w,
                            int64_t w_gstride, int64_t w_istride, const float* denom, const int32_t* img,
                            const int32_t* argmax, float* dX, int G, int N, int L, void* stream);

/* score head tail (gpn.py:54-57): z = <hid[g,:] * keep[g,:] * keep_scale, w2> + b2,
 * score = sigmoid(z), loss = mean BCE(score, target) with target[g] = g < G/2 (log clamped at
 * -100 like nn.BCELoss).  loss may be NULL.  bwd: dz, dhid (through the dropout mask), dw2, db2. */
int subgc_gpn_score_fwd(const float* hid, const uint8_t* keep, float keep_scale, const float* w2,
                        const float* b2, float* score, float* loss, int G, int H, void* stream);
int subgc_gpn_score_bwd(const float* hid, const uint8_t* keep, float keep_scale, const float* w2,
                        const float* score, const float* dloss, float* dhid, float* dw2, float* db2,
                        int G, int H, void* stream);

/* sub-graph NMS by node-set IoU (replaces gpn.py:108-150, O(M^2) python sets).
 * score [M], idx int64 [M,N] (row stride idx_stride), len int32 [M] (valid prefix length).
 * order: score descending, ties -> larger index first.  A sub-graph is dropped when an earlier
 * kept one has |A∩B|/|A∪B| > thres (evaluated in double like the reference).  The first
 * max_keep survivors are returned ASCENDING in original index: keep[0..*n_keep).  Node ids
 * index the image's N nodes and must be < 64*SUBGC_NMS_WORDS: N > 64*SUBGC_NMS_WORDS is
 * SUBGC_EINVAL (both forms).  scratch: M * (SUBGC_NMS_WORDS*8 + 8) bytes.                    */
#define SUBGC_NMS_WORDS 4
int subgc_subgraph_nms(const float* score, const int64_t* idx, int64_t idx_stride, const int32_t* len,
                       int M, int N, double thres, int max_keep, int64_t* keep, int32_t* n_keep,
                       void* scratch, size_t scratch_bytes, void* stream);
/* The same NMS for MANY images in one launch (one workgroup per image): image b owns candidates
 * offsets[b] .. offsets[b+1] (offsets int32 [images+1], total = offsets[images], max_m = the largest segment);
 * keep[offsets[b] + i] = i-th kept candidate of image b (index RELATIVE to its segment), n_keep[b] its count.
 * scratch: total * (SUBGC_NMS_WORDS*8 + 8) bytes.                                                              */
int subgc_subgraph_nms_batched(const float* score, const int64_t* idx, int64_t idx_stride, const int32_t* len,
                               const int32_t* offsets, int images, int total, int max_m, int N, double thres,
                               int max_keep, int64_t* keep, int32_t* n_keep, void* scratch, size_t scratch_bytes,
                               void* stream);

/* ======================================================================================
 * Decoder kernels
 */
/* ragged attention layout: off[s] = exclusive prefix sum of len[s] (len = int32 valid nodes of
 * sentence s), total -> *total.  Also emits for every packed row m in [off[s], off[s]+len[s]):
 * src_row[m] = img[s]*N + idx[s, m-off[s]]  (the X_out row feeding att_embed) and
 * sent_of[m] = s.  Rows m >= total get src_row = -1.  (replaces AttModel.py:348-354 clip_att and
 * :16-36 sort/pack/unpack.)  idx int64 [S,N] with row stride idx_stride.                      */
int subgc_pack_rows(const int32_t* len, const int64_t* idx, int64_t idx_stride, const int32_t* img,
                    int S, int N, int32_t* off, int32_t* total, int32_t* src_row, int32_t* sent_of,
                    void* stream);

/* xt = dropout(relu(Emb[tok])) for a block of tokens (AttModel.py:332 over all teacher-forced
 * steps at once).  tok int64 [n] (tok_stride elements apart); out [n,E].  bwd scatter-adds
 * dEmb[tok] += dxt * [Emb[tok] > 0] * keep * scale with fp32 atomics.                         */
int subgc_embed_fwd(const float* table, const int64_t* tok, int64_t tok_stride, const uint8_t* keep,
                    float keep_scale, void* out, int n, int E, int vocab_rows, int out_bf16, void* stream);
int subgc_embed_bwd(const float* table, const int64_t* tok, int64_t tok_stride, const uint8_t* keep,
                    float keep_scale, const float* dout, float* dtable, int n, int E, int vocab_rows,
                    void* stream);

/* One LSTMCell step (AttModel.py:411, :423 -> nn.LSTMCell) for a decode batch of S <= 32 rows in ONE launch: the gate
 * GEMM x[S,K] . w_perm[4R,K]^T streams the weights through the matrix pipe and the cell update runs in its epilogue.
 * w_perm is the K-concatenated gate matrix [W_ih | W_hh] with PERMUTED rows: row 16*b + 4*g + u holds gate g (i,f,g,o)
 * of hidden unit 4*b + u, so a workgroup's 16 rows are everything four units need.  pre = x.w^T + b0 + b1
 * + add1[tok ? tok[s] : s] + add2[s]  (add1/add2 [*,4R] in the UNpermuted gate order, optional; tok int64 [S] selects
 * rows of a [tok_rows,4R] table);  c = f*c_prev + i*g;  h = o*tanh(c) is stored to every non-null h0/h1/h2 (row
 * stride ldh*).  x must not alias any h destination (the caller ping-pongs its state buffers).  R % 4 == 0.          */
int subgc_lstm_step_skinny(const float* x, int64_t ldx, const void* w_perm, int64_t ldw, int K, int S, int R,
                           const float* add1, int64_t ld1, const int64_t* tok, int tok_rows, const float* add2,
                           int64_t ld2, const float* b0, const float* b1, const float* c_prev, float* c, float* h0,
                           int64_t ldh0, float* h1, int64_t ldh1, float* h2, int64_t ldh2, int w_bf16, void* stream);

/* out[r,:] = table[tok[r],:] -- plain token-row lookup, float4 when the rows allow it.  Decode only: with frozen weights
 * the x->gates product of the attention LSTM, relu(Emb) . W_ih[:, 2R:]^T (AttModel.py:332 feeding :409-411), depends on the
 * token alone, so the host builds that [V+1, 4R] table once per set of weights and each step looks rows up instead of
 * running an [n,E]x[E,4R] GEMM.  tok int64 [n] (tok_stride apart, clamped to [0, vocab_rows)); table [vocab_rows, C].   */
int subgc_token_rows_f32(const float* table, int64_t ldt, const int64_t* tok, int64_t tok_stride, float* out,
                         int64_t ldo, int n, int C, int vocab_rows, void* stream);

/* fused LSTM cell pointwise (nn.LSTMCell, gate order i,f,g,o; AttModel.py:413,423):
 *   pre = g0 + g1 + g2 + b0 + b1   (g1, g2, b0, b1 may be NULL; each g* is [S,4R] with its own ld)
 *   c = sig(f) c_prev + sig(i) tanh(g);  h = sig(o) tanh(c);  gates[S,4R] keeps the activated
 *   i,f,g,o for the backward (may be NULL at inference).  h is written with leading dim ldh so
 *   it can land inside the next GEMM's concatenated operand; h2 (optional) receives a second copy
 *   (ldh2), hdrop (optional) the dropout-masked copy used by the logit layer.  rows_h / rows_h2
 *   (0 = all S) limit how many leading rows of h / h2 are written: with sentences sorted by length
 *   the next step only consumes a prefix of them (packed decoder).                              */
int subgc_lstm_fwd(const float* g0, int64_t ld0, const float* g1, int64_t ld1, const float* g2, int64_t ld2,
                   const float* b0, const float* b1, const float* c_prev, float* c, void* h, int64_t ldh,
                   void* h2, int64_t ldh2, const uint8_t* keep, float keep_scale, void* hdrop, int64_t ldhd,
                   float* gates, int S, int R, int rows_h, int rows_h2, int h_bf16,
                   void* stream);
/* subgc_gemm_f32(x . w^T) + subgc_lstm_fwd in one call for the teacher-forced steps (S in the hundreds): when the product
 * takes the split-K form its partial planes stay in the workspace and the cell kernel adds them while it reads the
 * pre-activations (no reduce launch, no [S,4R] round trip); otherwise the product goes to `pre` [S, >= 4R] (scratch) and the
 * two entry points run back to back.  w [4R, K] = [W_ih(cols) | W_hh] K-concatenated; remaining arguments as subgc_lstm_fwd.
 * bf16_bits: bit 0 = x and w are bf16 (the product runs as subgc_gemm_bf16), bit 1 = the h destinations are bf16;
 * gemm_flags: the SUBGC_GEMM_MODE_* bits of the fp32 product; workspace / ws_bytes: as subgc_gemm_f32.                    */
int subgc_lstm_fwd_gemm(const void* x, int64_t ldx, const void* w, int64_t ldw, int K, float* pre, int64_t ldpre,
                        const float* g1, int64_t ld1, const float* g2, int64_t ld2, const float* b0, const float* b1,
                        const float* c_prev, float* c, void* h, int64_t ldh, void* h2, int64_t ldh2,
                        const uint8_t* keep, float keep_scale, void* hdrop, int64_t ldhd, float* gates, int S, int R,
                        int rows_h, int rows_h2, int bf16_bits, int gemm_flags, void* workspace, size_t ws_bytes,
                        void* stream);


/* dh (up to two sources summed: dh_a, dh_b, either may be NULL) and dc (may be NULL) ->
 * dpre [S,4R] and dc_prev.  dh_drop (optional) is a gradient that arrives through the dropout
 * mask (keep/keep_scale).                                                                    */
int subgc_lstm_bwd(const float* gates, const float* c_prev, const float* c, const float* dh_a, int64_t lda,
                   const float* dh_b, int64_t ldb, const float* dh_drop, int64_t ldd, const uint8_t* keep,
                   float keep_scale, const float* dc, void* dpre, float* dc_prev, int S, int R, int dpre_bf16,
                   void* stream);
/* Split-K partial planes consumed in place (the backward's recurrent data-gradient products, M <= a few hundred rows, are split-K;
 * their consumers read d(h) / d(ctx) exactly once, so they add the planes themselves and the reduce launches disappear):
 *   subgc_gemm_f32_planes / subgc_gemm_bf16_planes: op(A) op(B) as *n_planes fp32 planes planes[q][M][N] (stride M*N) whose sum is the
 *     product -- the dispatch's split-K form without its reduce pass; *n_planes = 1 when it does not split (plane 0 = the product).
 *   subgc_lstm_bwd_planes: subgc_lstm_bwd with up to three d(h) sources, source i = sum_{q < n_i} p_i[q*stride_i + s*ld_i + j] for rows
 *     s < rows_i (a column window of a plane stack: ld_i = the planes' N, p_i offset by the window's first column).
 *   subgc_attn_bwd_planes: subgc_attn_bwd with d(ctx) = sum of dctx_planes planes dctx + q * plane_stride.                      */
int subgc_gemm_f32_planes(int transA, int transB, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                          float* planes, size_t planes_bytes, int* n_planes, int flags, void* stream);
int subgc_gemm_bf16_planes(int transA, int transB, int M, int N, int K, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                           float* planes, size_t planes_bytes, int* n_planes, void* stream);
int subgc_lstm_bwd_planes(const float* gates, const float* c_prev, const float* c, const float* p0, int64_t ld0, int n0, int64_t stride0,
                          int rows0, const float* p1, int64_t ld1, int n1, int64_t stride1, int rows1, const float* p2, int64_t ld2, int n2,
                          int64_t stride2, int rows2, const float* dh_drop, int64_t ldd, const uint8_t* keep, float keep_scale,
                          const float* dc, void* dpre, float* dc_prev, int S, int R, int dpre_bf16, void* stream);
int subgc_attn_bwd_planes(const void* u, const void* v, const float* ah, const float* w_a, const int32_t* off, const int32_t* len,
                          const float* alpha, int n_stride, const float* dctx, int64_t lddctx, int dctx_planes, int64_t plane_stride,
                          void* dah, float* du, float* dv, float* dw_a, float* db_a, int S, int A, int R, int bf16_bits,
                          float* dctx_keep, int64_t ldkeep, void* stream);

/* one attention step over the ragged node sets (AttModel.py:453-466):
 *   e_i = <w_a, tanh(u[m,:] + ah[s,:])> + b_a ; alpha = softmax over the sentence's valid rows
 *   (== softmax-then-mask-then-renormalise of the reference up to rounding) ;
 *   ctx[s,:] = sum_i alpha_i v[m,:],  m = off[s]+i.
 * u [rows,A], v [rows,R] packed; ah [S,A]; alpha_out [S,n_stride] (entries i >= len are 0);
 * ctx written with leading dim ldctx.                                                        */
int subgc_attn_fwd(const void* u, const void* v, const float* ah, const float* w_a, const float* b_a,
                   const int32_t* off, const int32_t* len, void* ctx, int64_t ldctx, float* alpha,
                   int n_stride, int S, int A, int R, int bf16_bits, void* stream);
/* subgc_attn_fwd whose query rows are still the split-K partial planes of the h2att product (AttModel.py:455 `self.h2att(h)`;
 * subgc_gemm_*_planes): ah[s, :] = q_bias + sum_{p < n_planes} (q_planes + p * plane_stride)[s, :].  The summed rows are written to
 * q_out [S, A] -- the `ah` the backward takes.  Saves the reduce pass of that product (17 launches per train step). */
int subgc_attn_fwd_q(const void* u, const void* v, const float* q_planes, int n_planes, int64_t plane_stride, const float* q_bias,
                     float* q_out, const float* w_a, const float* b_a, const int32_t* off, const int32_t* len, void* ctx, int64_t ldctx,
                     float* alpha, int n_stride, int S, int A, int R, int bf16_bits, void* stream);
/* bf16_bits of subgc_attn_fwd / subgc_attn_bwd: bit 0 = the ctx (fwd) / dah (bwd) destination is bf16; bit 1 = u and v are
 * bf16 [rows, A] / [rows, R] (compute_dtype = bf16: the node features survive only as the tensors the GEMMs wrote).            */
/* backward of one step: dctx [S,R] (ld lddctx) -> dah [S,A]; du, dv ACCUMULATE (+=) over steps;
 * dw_a [S,A] and db_a [S] receive PER-SENTENCE partial gradients of w_a / b_a (plain stores; the
 * caller column-sums them once over all steps: 640 workgroups x 512 same-address atomics per step
 * cost more than the rest of the kernel).
 * Deferred d(v): with dv == NULL the step leaves d(v) alone and (dctx_keep != NULL) stores a copy of its d(ctx) rows to
 * dctx_keep [S, >= R] (ld ldkeep); after the time loop ONE call of subgc_attn_dv_accum forms d(v) from the kept rows and the
 * attention weights of all steps -- every d(v) row is then written once instead of read and written at every step.           */
int subgc_attn_bwd(const void* u, const void* v, const float* ah, const float* w_a, const int32_t* off,
                   const int32_t* len, const float* alpha, int n_stride, const float* dctx, int64_t lddctx,
                   void* dah, float* du, float* dv, float* dw_a, float* db_a, int S, int A, int R,
                   int bf16_bits, float* dctx_keep, int64_t ldkeep, void* stream);
/* Deferred d(u) (the same idea as the deferred d(v)): subgc_attn_bwd_planes_de is subgc_attn_bwd_planes that does NOT touch d(u) but
 * files the step's d(e) rows, de_keep [S, n_stride] (entries i >= len untouched); after the time loop ONE call of subgc_attn_du_accum
 * forms
 *   du[off[s] + i, a] = w_a[a] * sum over steps t < T with s < step_off[t+1] - step_off[t] of
 *                       de[step_off[t] + s, i] * (1 - tanh^2(u[off[s] + i, a] + ah[step_off[t] + s, a]))            (overwrites du)
 * from the kept d(e) rows and the query rows ah [rows, A] the forward saved (layout of de / ah: as alpha / dctx of
 * subgc_attn_dv_accum).  Every d(u) row is written once instead of read and written at every step (Full-GC: 36 node rows x 512 x 8 bytes
 * per sentence and step were the largest stream of the backward attention kernel); the price is one more tanh per (step, node, column).
 * u: fp32 or bf16 (uv_bf16) [sum len, A]; the accumulation order over t differs from the per-step form (fp32 either way).           */
int subgc_attn_bwd_planes_de(const void* u, const void* v, const float* ah, const float* w_a, const int32_t* off, const int32_t* len,
                             const float* alpha, int n_stride, const float* dctx, int64_t lddctx, int dctx_planes, int64_t plane_stride,
                             void* dah, float* de_keep, float* dv, float* dw_a, float* db_a, int S, int A, int R, int bf16_bits,
                             float* dctx_keep, int64_t ldkeep, void* stream);
int subgc_attn_du_accum(const void* u, int uv_bf16, const float* ah, const float* de, int n_stride, const int32_t* step_off, int T,
                        const int32_t* off, const int32_t* len, const float* w_a, float* du, int S, int A, void* stream);
/* dv[off[s] + i, :] = sum over steps t < T with s < step_off[t+1] - step_off[t] of
 *                     alpha[step_off[t] + s, i] * dctx[step_off[t] + s, :]              (overwrites dv)
 * alpha [rows, n_stride] and dctx [rows, >= R] (ld lddctx) hold the live sentences of step t as rows step_off[t] ..
 * step_off[t+1]-1 in sentence order (the packed decoder's layout; step_off[t] = t*S for the unpacked one); step_off: int32
 * [T+1] on the device.  off / len [S]: each sentence's node rows in dv [sum len, R].                                         */
int subgc_attn_dv_accum(const float* alpha, int n_stride, const float* dctx, int64_t lddctx, const int32_t* step_off,
                        int T, const int32_t* off, const int32_t* len, float* dv, int S, int R, void* stream);

/* in-place row log-softmax of logits[rows, V] (AttModel.py:336,340).  active (int32 [rows] or
 * NULL): rows with active == 0 are written as zeros (the reference leaves `outputs` rows of the
 * steps after its early break at zero, AttModel.py:152,171-172).                              */
int subgc_log_softmax_rows(float* x, int64_t ldx, int rows, int V, const int32_t* active, void* stream);
/* lse[r] = log sum_c exp(x[r, c]) (one read of the row, nothing written back): the loss-only path keeps RAW logits and hands `lse` to
 * subgc_masked_nll_fwd / subgc_nll_logsoftmax_bwd, which subtract it on the fly -- log_softmax (AttModel.py:336) without the write. */
int subgc_row_lse_f32(const float* x, int64_t ldx, int rows, int V, float* lse, void* stream);
/* dlogits = dout - exp(logp) * sum(dout); dlogits may alias dout (in place).                 */
int subgc_log_softmax_rows_bwd(const float* logp, const float* dout, void* dlogits, int64_t ld, int rows,
                               int V, const int32_t* active, int out_bf16, void* stream);
/* LanguageModelCriterion (misc/utils.py:115-124): num = -sum mask*logp[target], den = sum mask,
 * loss = num/den.  logp [S,T,V]; target, mask are [S,T] views of the [S,T+1] label/mask tensors
 * shifted by one (row strides t_stride / m_stride).  bwd writes dlogp (dense, zero elsewhere). */
/* den_override (device float*, may be NULL): use *den_override as the denominator instead of the mask sum of the rows given
 * (the packed decoder hands in only the live rows; the reference divides by the sum of ALL mask entries).
 * lse (float [S*T], may be NULL): `logp` holds raw logits, the log-probability of row q is logp[q, w] - lse[q]. */
int subgc_masked_nll_fwd(const float* logp, const int64_t* target, int64_t t_stride, const float* mask,
                         int64_t m_stride, float* loss, float* scratch2, int S, int T, int V, const float* den_override,
                         const float* lse, void* stream);
int subgc_masked_nll_bwd(const int64_t* target, int64_t t_stride, const float* mask, int64_t m_stride,
                         const float* scratch2, const float* dloss, float* dlogp, int S, int T, int V,
                         void* stream);
/* masked-NLL backward fused through the log-softmax (the criterion applied directly to the decoder's
 * log-probabilities, as LossWrapper does): dlogits = dloss * mask/den * (softmax - onehot(target));
 * never materialises the dense dlogp.  scratch2 is the {num, den} pair written by masked_nll_fwd.  lse (may be NULL): `logp` holds raw
 * logits and lse their row log-sum-exp (subgc_row_lse_f32).  */
int subgc_nll_logsoftmax_bwd(const float* logp, const int64_t* target, int64_t t_stride, const float* mask,
                             int64_t m_stride, const float* scratch2, const float* dloss, void* dlogits,
                             int64_t ld_out, int S, int T, int V, const int32_t* active, int out_bf16, const float* lse,
                             void* stream);
/* step_active[t] = 1 for t = 0 and for t >= 1 while no earlier step had all labels[:, t] == 0
 * (AttModel.py:171-172), expanded to rows: active[s*T + t].                                  */
int subgc_step_active(const int64_t* labels, int64_t l_stride, int S, int T, int32_t* active, void* stream);

/* ======================================================================================
 * Index plumbing of a training step on the device (csrc/plan.hip): one launch each instead of a dozen ATen ops.
 *
 * subgc_live_plan -- row plan of the packed (loss-only) decoder.  labels int64 [S, >= T] (row stride ld_labels), mask fp32
 *   [S, T] = the criterion mask masks[:, 1:] (row stride ld_mask).  live[s] = 1 + last t with mask[s,t] > 0, clipped to the
 *   reference's early break (AttModel.py:171-172: the loop stops at the first t >= 1 with labels[:, t] all zero).  perm32 /
 *   perm64 [S] (perm64 may be NULL): sentences by live steps, descending, stable.  counts[t] (t < T) = sentences with
 *   live > t; offs[t] (t <= T) = prefix of counts; den[0] = sum of ALL mask entries (misc/utils.py:123); inv32 (may be NULL):
 *   inverse permutation (inv32[perm32[j]] = j).  S <= 16384, T <= 63.
 * subgc_packed_rows -- with that plan (device arrays), packed row r = offs[t] + j <-> (sentence perm[j], step t):
 *   tok_flat[r] = labels[perm[j], t], tgt_p[r] = target[perm[j], t], msk_p[r] = mask[perm[j], t]; and the per-sentence inputs
 *   in sorted order: labels_p [S, label_cols], lens_p [S], idx_p [S, N] (from idx, row stride ld_idx), img_p [S].
 * subgc_gpn_prep -- gpn.py:43-52: the loader's [b5, 2, hb, ...] sub-graph tensors as the pos half then the neg half,
 *   g = c*(b5*hb) + s*hb + h: idx [G, N] node lists, w [G, N] = diagonal of gpn_pool_mtx, denom [G] = mask sums, img [G] =
 *   s / sentences_per_image.
 * subgc_gpn_select -- gpn.py:63-78: per sentence the FIRST max of its hb positive scores (score [2*b5*hb]), that sub-graph's
 *   node list sel_idx [b5, N], its node count lens [b5], its read-out row ro_sel [b5, read_out_cols] (ro_sel may be NULL) and
 *   the chosen slot sel [b5] (may be NULL); img_s [b5] (may be NULL) = owning image s / sentences_per_image.
 * subgc_add_n_f32 -- out = a + b (+ c) (+ d), n elements (out may alias a): the sum of the gradient contributions of a tensor with several consumers. */
int subgc_live_plan(const int64_t* labels, int64_t ld_labels, const float* mask, int64_t ld_mask, int S, int T, int32_t* perm32,
                    int64_t* perm64, int32_t* inv32, int32_t* counts, int32_t* offs, float* den, void* stream);
int subgc_packed_rows(const int64_t* labels, int64_t ld_labels, const int64_t* target, int64_t ld_target, const float* mask,
                      int64_t ld_mask, const int32_t* perm, const int32_t* offs, int S, int T, int64_t* labels_p, int label_cols,
                      int64_t* tok_flat, int64_t* tgt_p, float* msk_p, const int32_t* lens, const int64_t* idx, int64_t ld_idx,
                      const int32_t* img, int N, int32_t* lens_p, int64_t* idx_p, int32_t* img_p, void* stream);
int subgc_gpn_prep(const int64_t* gpn_obj_ind, const float* gpn_pool_mtx, const float* att_masks, int b5, int hb, int N,
                   int sentences_per_image, int64_t* idx, float* w, float* denom, int32_t* img, void* stream);
/* The survivors of subgc_subgraph_nms_batched in image order (total = sum of n_keep, known to the caller from its one host read):
 * keep[pos] = the kept candidate's index inside its image, glob[pos] = its row in the concatenated candidate arrays.                  */
int subgc_nms_compact(const int64_t* keep_all, const int32_t* n_keep, const int32_t* offsets, int images, int total, int64_t* keep,
                      int64_t* glob, void* stream);
/* out[b][0 .. words) = the first `words` 4-byte words of the tensor at device address ptrs[b] (ptrs: device int64 [count]): block 0
 * (counterpart 0) of every image's loader tensor stacked into one batch array -- the batched decode's input assembly
 * (dataloader_test.py hands one image per item, eval_utils.py:98-104 loops over them).                                                */
int subgc_gather_blocks(const int64_t* ptrs, int count, int64_t words, float* out, void* stream);
/* Per-image early break of a BATCHED decode (the reference decodes one image per call and stops when none of ITS rows is unfinished,
 * AttModel.py:318-319): image b = rows bounds[b] .. bounds[b+1] of seq [n, T] (int64) / seqlp [n, T]; log-probs after the image's break
 * step are zeroed in place; out[2b] = break step (T-1 when the image never stopped), out[2b+1] = 1 when some step had no unfinished row. */
int subgc_decode_batch_finish(const int64_t* seq, float* seqlp, const int32_t* bounds, int images, int T, int32_t* out, void* stream);
/* The sGPN TEST branch's input views (gpn.py:84-96 reads counterpart 0 of the loader's five identical copies) for MANY images in one
 * launch.  table (device, int64): [images + 1] candidate offsets (image b owns candidates offsets[b] .. offsets[b+1]), then 4 words per
 * image: the device addresses of its gpn_obj_ind [5, 2, M_b, N] (int64), gpn_pool_mtx [5, 2, M_b, N, N] and att_masks [5, 2, M_b, N]
 * tensors (contiguous) and its row in the node-state array.  -> idx [total, N] node lists, w [total, N] pooling weights (the diagonal of
 * gpn_pool_mtx), denom [total] / lens [total] node counts (float / int32), img [total] owning row, offsets32 [images + 1] (optional:
 * the offsets as int32, what subgc_subgraph_nms_batched takes).                                                                        */
int subgc_gpn_test_prep(const int64_t* table, int images, int total, int N, int64_t* idx, float* w, float* denom, int32_t* lens,
                        int32_t* img, int32_t* offsets32, void* stream);
int subgc_gpn_select(const float* score, const int64_t* gpn_obj_ind, const float* att_masks, const float* read_out, int b5, int hb,
                     int N, int read_out_cols, int64_t* sel_idx, int32_t* lens, float* ro_sel, int32_t* sel,
                     int sentences_per_image, int32_t* img_s, void* stream);
/* part[slab][c][:] = sum of the rows of X [M, L] (row stride ldx) whose class id cls[r] is c, per slab of ceil(M / slabs) rows
 * (part: slabs * C * L floats, C <= 64): the gradient of a class-indexed table (AttModel.py:383-386: the relation embedding
 * Emb_pred[argmax] followed by pred_emb_prj has only sg_pred_cnt = 21 distinct rows).  A column sum over the slabs finishes it. */
int subgc_class_partials(const float* X, int64_t ldx, const int32_t* cls, int M, int L, int C, int slabs, float* part, void* stream);
int subgc_add_n_f32(float* out, const float* a, const float* b, const float* c, const float* d, int64_t n, void* stream);
/* x[r, c] = value over a [rows, cols] window (row stride ld): AttModel.py:148-149 `att_masks[:, :36] = 1` on the caller's tensor */
int subgc_fill2d_f32(float* x, int64_t ld, int rows, int cols, float value, void* stream);
/* lens[r] = (int32) sum of row r of a 0/1 mask: node counts of the attention sets (AttModel.py:348-354 clip_att's mask sums) */
int subgc_row_count_f32(const float* x, int64_t ld, int rows, int cols, int32_t* lens, void* stream);

/* ======================================================================================
 * Greedy pick folded into the decode step's launches (csrc/gemm_skinny.hip; AttModel.py:295-319 with sample_max, <= 16 rows).
 * `best` buffers: uint64 [16 rows][8 slots][16] (16 KB; slot (m, x) at element (m*8 + x)*16, its own 128-byte line), all zero
 * before the logits launch that fills them.
 *   subgc_skinny_dual with `best` -- problem 1 = the logits x W^T + bias WITHOUT writing them (C1 may be NULL): slot (m, workgroup & 7) <-
 *       atomicMax of (ordered logit bits << 32 | ~column): the row's arg-max with ties to the smaller column (torch.max's first max);
 *       lse_part[(wg*16 + m)*2 + {0,1}] = (max, sum exp(. - max)) over the 16 vocabulary rows of workgroup wg (ceil(V/16) of them).
 *   subgc_lstm_cell_pick -- the attention LSTM's cell whose input word of row m is that arg-max (finished rows feed 0; the word selects
 *       row add1[word] of the per-token x->gates table); workgroup 0 files the pick of step t_prev exactly once: seq[m, t_prev],
 *       unf_out[m] (= unf_in[m] && word > 0; word > 0 at t_prev = 0), count_out += live rows; nothing is written when *prev_count
 *       == 0 (the reference has left its loop); `best_reset` (the other buffer) is cleared for this step's logits launch (NULL: none).
 *   subgc_pick_file    -- that bookkeeping alone (after the last pick).
 *   subgc_pick_lse_finish -- seqlp[m, t] = -log sum_wg sum_wg exp(max_wg - max) for every step the loop reached, from
 *       lse_part [T][ceil(V/16)][16][2]: one pass after the loop instead of a vocabulary reduction per step.
 * (Rounds 3-5 had the pick in a logits-only launch, subgc_logits_pick, and in the fused attention-LSTM launch, subgc_lstm_step_pick;
 * round 6 replaced both by the two entry points below.)                                                                          */
/* C[M,N] = act(A[M,K] W[N,K]^T + bias), M <= 16, with bf16-STORED W (raw uint16) and fp32 activations / results: the weight-streaming
 * products of a decode step under compute_dtype = bf16 (w_bf16 != 0 in subgc_lstm_step_skinny / subgc_skinny_dual selects the same for them:
 * half the bytes of the 120 MB a token step streams; accumulation and everything downstream stay fp32). */
int subgc_gemm_skinny_wb16(const float* A, int64_t lda, const uint16_t* W, int64_t ldw, float* C, int64_t ldc, const float* bias, int M,
                           int N, int K, int relu, void* stream);
int subgc_pick_file(const uint64_t* best_prev, const int32_t* unf_in, int32_t* unf_out, int64_t* seq, int S, int T, int t_prev,
                    int32_t* count_out, const int32_t* prev_count, void* stream);
/* Round 6 -- the independent weight streams of a one-image token step share launches (the dependent chain keeps five launches, the
 * attention LSTM's 32 MB and two thirds of the language LSTM's 48 MB leave it):
 * subgc_skinny_dual: two weight-streaming products of S <= 16 rows in ONE launch -- problem 1 with the fused arg-max / log-sum-exp
 *   epilogue described above when `best` != NULL (C1 may then be NULL), problem 2 plain.  unperm*_R != 0: that problem's W rows are in
 *   the LSTM forms' permuted gate order and its result is written gate-major [S, 4R].  The decode step uses it for
 *   [logits(h_lang) | H1 . Wc1^T] and for [h2att(h_att) | [h_att, h_lang_prev] . Wc2[:, R:]^T]  (AttModel.py:411-413, 421-423, 453, 336).
 * subgc_lstm_cell_pick: the attention LSTM's cell from those pre-activations + the picked word's x -> gates table row (add1) + the fc term
 *   (add2) + biases; the word is the previous step's fused arg-max (best_prev; the pick is filed as described above) or tok[m].          */
int subgc_skinny_dual(int S, const float* x1, int64_t ldx1, const void* W1, int64_t ldw1, const float* bias1, int N1, int K1, float* C1,
                      int64_t ldc1, int unperm1_R, uint64_t* best, float* lse_part, const float* x2, int64_t ldx2, const void* W2,
                      int64_t ldw2, const float* bias2, int N2, int K2, float* C2, int64_t ldc2, int unperm2_R, int w_bf16, void* stream);
int subgc_lstm_cell_pick(const float* pre, int64_t ldpre, int S, int R, const float* add1, int64_t ld1, const int64_t* tok, int tok_rows,
                         const float* add2, int64_t ld2, const float* b0, const float* b1, const float* c_prev, float* c, float* h0,
                         int64_t ldh0, float* h1, int64_t ldh1, float* h2, int64_t ldh2, const uint64_t* best_prev, const int32_t* unf_in,
                         int32_t* unf_out, int64_t* seq, int T, int t_prev, int32_t* count_out, const int32_t* prev_count,
                         uint64_t* best_reset, void* stream);
int subgc_pick_lse_finish(const float* lse_part, int V, int S, int T, const int32_t* counts, float* seqlp, void* stream);

/* ======================================================================================
 * Attention over SHARED sets (csrc/attention_group.hip): the Full-GC model attends, for each of an image's g sentences, over the same
 * node rows (AttModel.py:140-149; the reference replicates them g = 5 times, gcn_backbone.py:50-51).  u [B*Nn, A], v [B*Nn, R] exist
 * once per image (fp32 or bf16: bf16_bits bit 1), one workgroup per image serves all its live sentences.
 *   rows int32 [B*g]: position of sentence j of image b in the step's row arrays (ah / ctx / alpha / dctx ... are indexed by it);
 *   the sentence takes part iff 0 <= rows[b*g+j] < m.  lens int32 [by row]: valid nodes (<= Nn <= 128).  g <= 8.
 *   fwd: ctx[row] (fp32 / bf16: bit 0), alpha[row, 0..n_stride).  bwd: dah[row] (fp32 / bf16: bit 0), du [B*Nn, A] += (zero it
 *   before the first step), dw_a[row, A] / db_a[row] per-sentence partials, dctx_keep[row] (may be NULL) = this step's d(ctx) rows;
 *   d(ctx) = sum of dctx_planes planes dctx + q * plane_stride (1, 0: a plain array).  An image is served by
 *   subgc_attn_group_du_planes(g) workgroups (1 or 2: a second wave per SIMD hides the tanh chains' latency); each accumulates into
 *   its own d(u) plane du + p * du_plane_stride -- the caller zeroes du_planes planes and adds them after the last step.
 *   dv_accum: dv [B*Nn, R] = sum over steps t and live sentences of alpha_t[row, i] * dctx_t[row, :]; step t's rows start at
 *   step_off[t] with step_off[t+1] - step_off[t] of them live (every row of dv is written).                                      */
int subgc_attn_fwd_group(const void* u, const void* v, const float* ah, const float* w_a, const float* b_a, const int32_t* rows,
                         const int32_t* lens, int m, int B, int g, int Nn, void* ctx, int64_t ldctx, float* alpha, int n_stride, int A,
                         int R, int bf16_bits, void* stream);
/* subgc_attn_fwd_group with the query rows as partial planes (see subgc_attn_fwd_q); q_out [m, A]. */
int subgc_attn_fwd_group_q(const void* u, const void* v, const float* q_planes, int n_planes, int64_t plane_stride, const float* q_bias,
                           float* q_out, const float* w_a, const float* b_a, const int32_t* rows, const int32_t* lens, int m, int B, int g,
                           int Nn, void* ctx, int64_t ldctx, float* alpha, int n_stride, int A, int R, int bf16_bits, void* stream);
int subgc_attn_bwd_group(const void* u, const void* v, const float* ah, const float* w_a, const int32_t* rows, const int32_t* lens, int m,
                         int B, int g, int Nn, const float* alpha, int n_stride, const float* dctx, int64_t lddctx, void* dah, float* du,
                         float* dw_a, float* db_a, int A, int R, int bf16_bits, float* dctx_keep, int64_t ldkeep, int dctx_planes,
                         int64_t plane_stride, int du_planes, int64_t du_plane_stride, void* stream);
int subgc_attn_group_du_planes(int g);
int subgc_attn_dv_accum_group(const float* alpha, int n_stride, const float* dctx, int64_t lddctx, const int32_t* step_off, int T,
                              const int32_t* rows, int B, int g, int Nn, float* dv, int R, void* stream);

/* greedy / top-k token choice of one decode step (AttModel.py:295-316).
 * greedy (k == 0): it = first argmax, lp = max.  top-k: lp' = log_softmax(logp/temp), keep the
 * k largest (ties -> smaller index), renormalise, draw by inverse CDF with uniform u[s];
 * lp = lp'[it].  Then unfinished &= it > 0; it *= unfinished; seq[s, t] = it; seqlp[s, t] = lp
 * (un-masked, like the reference); next_tok[s] = it; *n_unfinished (zeroed by the caller) becomes
 * NON-ZERO iff some row is still unfinished (a flag: plain stores, no same-address atomics).
 * prev_count (device int32*, may be NULL): that flag after the previous step; when it is 0
 * the kernel writes nothing -- the reference has broken out of its loop (AttModel.py:318-319) --
 * so the whole decode loop runs without a host round trip.
 * raw_logits != 0: `logp` holds un-normalised logits; the kernel folds the log-softmax in (the greedy
 * log-prob is -log sum exp(x - max); the top-k branch is shift-invariant), so decode never writes the
 * normalised [n, V+1] row.                                                                      */
int subgc_decode_pick(const float* logp, int64_t ld, int n, int V, int k, float temp, const float* u,
                      int t, int64_t* seq, float* seqlp, int T, int64_t* next_tok, int32_t* unfinished,
                      int32_t* n_unfinished, const int32_t* prev_count, int raw_logits, void* stream);

/* Beam search (CaptionModel.py:60 `torch.sort(logprobsf, 1, True)`, of which beam_step :62-72 reads only the
 * leading `beam` columns): vals[r, :k], idx[r, :k] = the k largest entries of row r of x[rows, cols] in
 * (value descending, index ascending) order.  log_softmax != 0: x holds raw logits and vals are
 * x - logsumexp(x) (AttModel.py:340 folded in).  k <= 32, cols <= 16384.                          */
int subgc_row_topk_f32(const float* x, int64_t ld, int rows, int cols, int k, int log_softmax, float* vals,
                       int32_t* idx, void* stream);

/* One step of (diverse) beam search for n sub-graphs at once (CaptionModel.py:28-94 beam_step + add_diversity, :126-166 the
 * per-group loop body), one workgroup per sub-graph, the groups of a sub-graph in order.  Rows are laid out
 * [n][G][bd] (bd = beam_size / group_size beams per group).  tv/ti [n][G][bd][kk]: the kk leading log-probs of every beam
 * row in (value desc, index asc) order and their word ids (subgc_row_topk_f32; kk = beam+2 suffices, see subgc/beam.py).
 * For every group that is live at global step t (g <= t <= T+g-1, local step tau = t-g) the kernel applies the
 * decoding constraint (:134-135), the UNK penalty -1000 (:137) and the diversity penalty lam per earlier-group pick
 * (:33-40), orders each row (stable, descending), forms the candidate list in (column, beam) order with fp32 sums
 * (:62-72), keeps the leading bd of its stable sort, forks the token / log-prob history seq, lps [n][G][T][bd] and the
 * running sums [n][G][bd] (:76-90), appends finished beams (word 0, or the last step) to done_* in finishing order
 * (done_cnt [n][G]; done_seq/done_lps [n][G][cap][T]; done_p = the sum BEFORE the length penalty; done_len = tau+1) and
 * sets their sum to -1000 (:150-166).  Outputs for the next decoder step: tok [n][G][bd] (0 for sleeping groups) and
 * src [n][G][bd], the state row every slot continues from.  T <= 64, bd <= 16, kk <= 18.                          */
int subgc_beam_step(const float* tv, const int32_t* ti, int32_t* seq, float* lps, float* sums, int32_t* done_cnt,
                    int32_t* done_seq, float* done_lps, float* done_p, int32_t* done_len, int64_t* tok,
                    int32_t* src, int n, int t, int T, int G, int bd, int kk, int unk, int constraint, float lam,
                    int cap, void* stream);

/* Eval loop (misc/eval_utils.py:106-108): order[r] = index of the r-th largest score (ties keep input order),
 * sorted[r] = that score (may be NULL).  n <= 8192 (an image has at most 2M candidate sub-graphs).     */
int subgc_rank_desc_f32(const float* score, int n, int64_t* order, float* sorted, void* stream);

/* The same ranking for MANY images in one launch (misc/eval_utils.py:105-115, the body of the one-image-per-call loop): image i
 * owns rows seg[i] .. seg[i+1]-1 (<= max_rows <= 8192 each) of a decode batch.  order[seg[i]+r] = image-LOCAL index of its r-th best
 * row (score descending, ties keep input order = subgc_rank_desc_f32); score_sorted / keep_sorted / seq_sorted [rows, T] are the
 * image's scores, kept sub-graph indices (`keep_nms_ind[sort_ind]`) and token rows (`seqq[sort_ind]`) in that order, narrowed to
 * int32.  identity != 0: no sorting (Full-GC, :112-115: sort_ind = arange).                                                   */
int subgc_eval_rank_rows(const float* score, const int64_t* keep, const int64_t* seq, int T, const int32_t* seg, int I,
                         int max_rows, int identity, int32_t* order, float* score_sorted, int32_t* keep_sorted,
                         int32_t* seq_sorted, void* stream);

/* Grounding material, integer part (misc/grd_utils.py:36-47, collected per image at misc/eval_utils.py:143-146).  For image i the
 * chosen caption is row g = seg[i] + order[seg[i] + pick[i]] of the decode batch (pick NULL: 0 = the best-ranked sentence; a
 * consensus re-ranker passes its own subg_index; order NULL: identity, the Full-GC branch :44-46).  n_words[i] = the words
 * decode_sequence emits for it = tokens of seq[g, :T] before the first 0.  For word position j < n_words[i]:
 *   att2[i, j] = FIRST arg-max over the N columns of AL[j, g, :]   (`torch.max(att_weights[row], dim=1)[1][:len(words)]`; AL is the
 *                decode loop's time-major attention buffer [T1, rows, N], strides ld_t / ld_row in elements -- columns past the
 *                image's n_max are zero, so the arg-max over N columns equals the one over the clipped tensor),
 *   node[i, j] = idx[g, att2[i, j]]   (`obj_ind_this[att2_ind[wd_j]]`: the sub-graph's node list -> full-graph node id = box row;
 *                Full-GC: idx rows are arange).
 * Positions j >= n_words[i] hold -1.  att2, node: int32 [I, T1]; T <= 64.                                                      */
int subgc_grounding_argmax(const float* AL, int64_t ld_t, int64_t ld_row, int N, int T1, const int64_t* seq, int T,
                           const int64_t* idx, int64_t ld_idx, const int32_t* seg, const int32_t* order, const int32_t* pick,
                           int I, int32_t* att2, int32_t* node, int32_t* n_words, void* stream);

/* ---- bf16-operand GEMM (BASELINE configs 3 / 5: "bf16") ---------------------------------------------------------------
 * C = epilogue(op(A) . op(B)) with A, B STORED as bf16 (raw uint16 bit patterns), fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16; the result goes to C32 (fp32) and / or C16 (bf16, round-to-nearest-even) -- either may be NULL.
 * Forms: (transA, transB) = (0,1) x[M,K] W[N,K]^T | (0,0) x[M,K] W[K,N] | (1,0) A stored [K,M], B [K,N].  Bases 16-byte
 * aligned, lda / ldb multiples of 8 (M, N, K arbitrary: a row may be over-read up to its ld, which is masked).
 * Epilogue as subgc_gemm_f32 (bias[N], add[M,N] fp32, ReLU, dense keep[M,N] mask x keep_scale, ACCUM into C32); m_dev bounds
 * the rows of the stored A.  `workspace` / `ws_bytes`: caller-owned scratch for the split-K form of THIS call (may be NULL:
 * no split); stream-ordered, so calls on different streams need different workspaces.                                  */
int subgc_gemm_bf16(int transA, int transB, int M, int N, int K, const uint16_t* A, int64_t lda, const uint16_t* B,
                    int64_t ldb, float* C32, int64_t ldc32, uint16_t* C16, int64_t ldc16, const float* bias,
                    const float* add, int64_t ldadd, const uint8_t* keep, float keep_scale, int flags,
                    const int32_t* m_dev, void* workspace, size_t ws_bytes, void* stream);
int subgc_gemm_bf16_workspace_bytes(int M, int N, int K, size_t* bytes);
/* Two subgc_gemm_bf16 products of the SAME shape, layout, leading dimensions and epilogue in ONE launch: (A1, B1 -> C*_1) in the first half
 * of the grid, (A2, B2 -> C*_2) in the second.  For the two collection units of a GCN pair (models/lib/graph_conv.py:24-25,31-32; each
 * unit graph_conv_unit.py:28-30): their d(H) = dy W_rgt halves and their fc_rgt weight gradients are half-filling launches one by one;
 * tile choice, K parts and row cut are planned for both together.  Epilogue: bias1 / bias2, SUBGC_GEMM_RELU, SUBGC_GEMM_ACCUM; either
 * kind of destination may be absent (for both problems alike).                                                                        */
int subgc_gemm_bf16_pair(int transA, int transB, int M, int N, int K, const uint16_t* A1, const uint16_t* A2, int64_t lda,
                         const uint16_t* B1, const uint16_t* B2, int64_t ldb, float* C32_1, float* C32_2, int64_t ldc32, uint16_t* C16_1,
                         uint16_t* C16_2, int64_t ldc16, const float* bias1, const float* bias2, int flags, void* workspace, size_t ws_bytes,
                         void* stream);
/* subgc_gemm_f32_wgrad over bf16-STORED dY [K, M] and X [K, N]: dW[M, N] (+)= dY^T X and db[M] (+)= sum_k dY[k, :], both fp32; the column
 * sums are read from the tile images of dY already in LDS (workgroups of tile column 0), fixed summation order.  Alignment rules of
 * subgc_gemm_bf16 (bases 16 bytes, lddy / ldx multiples of 8); flags: SUBGC_GEMM_ACCUM for dW, db_accumulate for db.                 */
int subgc_gemm_bf16_wgrad(int M, int N, int K, const uint16_t* dY, int64_t lddy, const uint16_t* X, int64_t ldx, float* dW, int64_t lddw,
                          float* db, int flags, int db_accumulate, const int32_t* m_dev, void* workspace, size_t ws_bytes, void* stream);
/* y[r, :cols_pad] = bf16(x[r, :cols]) with zero padding columns (cols_pad % 8 == 0, ldy % 8 == 0); rows bounded by *m_dev */
int subgc_cast_f32_bf16(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int rows, int cols, int cols_pad,
                        const int32_t* m_dev, void* stream);
/* y[c, r] = bf16(x[r, c]) (the W^T snapshots: every data-gradient product becomes an x W^T one) */
int subgc_transpose_f32_bf16(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int rows, int cols, void* stream);
int subgc_copy2d_b16(const uint16_t* x, int64_t ldx, uint16_t* y, int64_t ldy, int rows, int cols, void* stream);

/* ---- on-device batch assembly (dataloaders/dataloader.py:269-367) ----------------------------------------
 * mask_compact (:276-308): row g of the 0/1 `mask [G, W]` -> ind[g, :] = ascending set positions, padded with `pad`
 * (the dummy node / predicate index) to N columns; att_mask[g, i] = i < count (may be NULL); pool_mtx[g] = the
 * N x N diagonal 0/1 matrix of :283,290 (may be NULL -- the model only reads its diagonal).  W <= 1024.          */
int subgc_mask_compact(const uint8_t* mask, int64_t ld, int G, int W, int N, int64_t pad, int64_t* ind,
                       float* att_mask, float* pool_mtx, void* stream);
/* pad_rows (:336-354): image b owns rows off[b]..off[b+1] of the packed `src [sum, C]`; dst[b, r, :] = its r-th row
 * for r < min(count, limit), else the padding row: one-hot(0) (onehot0 != 0; the class-0 rows of :341,350) or zeros
 * (f32), `pad` (i64: the dummy endpoint obj_num-1 of :349).                                                       */
int subgc_pad_rows_f32(const float* src, const int64_t* off, int B, int R, int C, int limit, int onehot0,
                       float* dst, void* stream);
int subgc_pad_rows_i64(const int64_t* src, const int64_t* off, int B, int R, int C, int limit, int64_t pad,
                       int64_t* dst, void* stream);
/* pad_segments (:303-308, gpn_nrel_ind): group g owns rows start[g] .. start[g] + count[g] of the packed `src [sum, C]`
 * (segments in any order, repeats allowed); dst[g, r, :] = its r-th row for r < min(count[g], R), else `pad`.        */
int subgc_pad_segments_i64(const int64_t* src, const int64_t* start, const int64_t* count, int G, int R, int C,
                           int64_t pad, int64_t* dst, void* stream);
/* caption_labels (:356-363): labels[s] = [0, captions[s, :seq_length], 0]; masks[s, j] = j < nonzero(captions[s]) + 2 */
int subgc_caption_labels(const int64_t* captions, int64_t ld, int S, int seq_length, int64_t* labels,
                         float* masks, void* stream);

/* Scheduled sampling (AttModel.py:157-167).  uniform: out[i] = uniform(seed, offset + i) in [0, 1), the same
 * counter-based stream family as the dropout masks.  multinomial_rows: every row r with sel_u[r] < prob gets
 * tok[r * tok_stride] = inverse-CDF draw (index order, u[r]) from softmax(logits[r, :V]); other rows keep their word.
 * `logits` may be raw or log-normalised (the draw is shift-invariant).                                              */
int subgc_uniform_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
int subgc_multinomial_rows(const float* logits, int64_t ld, int rows, int V, const float* u, const float* sel_u,
                           float prob, int64_t* tok, int64_t tok_stride, void* stream);

/* Packed (length-sorted) decoder: dst[s, :C] = sum_t src[offsets[t] + s, :C] over the steps t with s < offsets[t+1] - offsets[t]
 * (offsets int32 [T+1] on the device, step sizes non-increasing).  The gradient of the loop-invariant fc->gates term. */
int subgc_packed_time_sum(const void* src, const int32_t* offsets, int T, int S, int C, float* dst, int src_bf16,
                          void* stream);

/* dropout keep-mask generator (counter-based, Philox-4x32-10): keep[i] = uniform(seed, offset+i) >= p */
int subgc_dropout_mask(uint8_t* keep, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);

/* small utilities */
/* dz = dy * scale * [y > 0]   (backward of y = relu(z) * keep * scale: y > 0 <=> z > 0 and kept)
 * bf16_bits: bit 0 = dz is bf16, bit 1 = y is bf16                                                */
int subgc_relu_bwd(const float* dy, const void* y, float scale, void* dz, int64_t n, int bf16_bits, void* stream);
/* dst[m, :] = src[rows[m], :] for m < min(M, *m_dev); negative rows give zero rows               */
int subgc_gather_rows(const float* src, int64_t lds, const int32_t* rows, void* dst, int64_t ldd,
                      int M, int L, const int32_t* m_dev, int out_bf16, void* stream);
/* dst[m, :] = src[rows[m], :] * (keep ? keep[m, :] * scale : 1) for m < min(M, *m_dev): nn.Dropout applied to GATHERED copies of shared
 * rows, every copy with its own keep-mask row (AttModel.py:113-119 on the x5 replicated node rows of gcn_backbone.py:50-51: one
 * att_embed product over the unique rows serves all copies).  bf16_bits: bit 0 = dst is bf16, bit 1 = src is bf16; keep may be NULL. */
int subgc_gather_rows_keep(const void* src, int64_t lds, const int32_t* rows, const uint8_t* keep, int64_t ldk, float scale,
                           void* dst, int64_t ldd, int M, int L, const int32_t* m_dev, int bf16_bits, void* stream);
/* the same gather for `count` (1..4) tensors in one launch: dst_k[m, :c_k] = src_k[rows[m], :c_k] -- the state fork of beam
 * search (CaptionModel.py:76-90: h and c of both LSTMs follow the surviving beams), four tensors per step            */
int subgc_gather_rows_multi(int count, const float* s0, int64_t lds0, float* d0, int64_t ldd0, int c0, const float* s1,
                            int64_t lds1, float* d1, int64_t ldd1, int c1, const float* s2, int64_t lds2, float* d2,
                            int64_t ldd2, int c2, const float* s3, int64_t lds3, float* d3, int64_t ldd3, int c3,
                            const int32_t* rows, int M, void* stream);
/* ... and with int64 row ids (torch's index dtype; the NMS survivor list): columns are 4-byte words, so int64 / int32 tensors pass as
 * their float32 views -- the decode-time selection of the kept sub-graphs' read-out rows, node lists, node counts and scores          */
int subgc_gather_rows_multi_i64(int count, const float* s0, int64_t lds0, float* d0, int64_t ldd0, int c0, const float* s1,
                                int64_t lds1, float* d1, int64_t ldd1, int c1, const float* s2, int64_t lds2, float* d2,
                                int64_t ldd2, int c2, const float* s3, int64_t lds3, float* d3, int64_t ldd3, int c3,
                                const int64_t* rows, int M, void* stream);
int subgc_fill_f32(float* x, int64_t n, float value, void* stream);
/* y[rows, cols] (ldy) += / = x[rows, cols] (ldx) */
int subgc_copy2d_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int cols, int accumulate,
                     void* stream);
/* dX[rows[m], :] += src[m, :] for m < min(M, *m_dev) (fp32 atomics; negative rows skipped) */
int subgc_scatter_add_rows(const float* src, int64_t lds, const int32_t* rows, float* dX, int64_t ldx,
                           int M, int L, const int32_t* m_dev, void* stream);

/* fused global-norm clip + Adam over one flat fp32 bucket (misc/utils.py:174-200,234-235):
 * pass 1 accumulates sum(g^2) into *sumsq (caller zeroes it), pass 2 applies
 * g *= max_norm / max(sqrt(sumsq), max_norm) (utils.py:193) and the torch.optim.Adam update.  grad_scale (1 for one GPU,
 * 1/world after a SUM all-reduce: DataParallel's mean of the replica losses, train.py:154-156) multiplies g first -- norm and
 * update see the averaged gradient without a separate pass over the bucket; g is left scaled and clipped.
 * p_bf16 (optional, n elements): the bf16 snapshot of the updated weights, written in the same sweep (what the bf16-storage
 * GEMMs of BASELINE configs 3 / 5 read).                                                                                */
int subgc_sumsq_f32(const float* g, int64_t n, float* sumsq, void* stream);
int subgc_clip_adam_step(float* p, float* g, float* m, float* v, int64_t n, const float* sumsq,
                         float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                         int step, float grad_scale, uint16_t* p_bf16, void* stream);
/* the same sweep with `optimizer.zero_grad()` (train.py calls it once per iteration) folded in: g is left ZEROED
 * instead of scaled and clipped, so the next iteration needs no fill pass over the gradient buffer.             */
int subgc_clip_adam_step_zero(float* p, float* g, float* m, float* v, int64_t n, const float* sumsq,
                              float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                              int step, float grad_scale, uint16_t* p_bf16, void* stream);

/* ======================================================================================
 * The teacher-forced recurrence as ONE call per direction (AttModel.py:157-175: the T-step loop around TopDownCore, :400-431).
 * The per-step launch groups -- att-LSTM product + cell, h2att product (kept as planes), attention, lang-LSTM product + cell;
 * backward: the two cell backwards, the attention backward and the three data-gradient products (kept as planes) -- are exactly
 * the entry points above, issued from C with pointer arithmetic instead of from the host language once per step: ~10 library
 * crossings per step x 17 steps become 2 per train step.  No kernel, no launch order and no argument differs from the
 * step-by-step sequence (tests: packed == unpacked == goldens).
 *
 * Step t (0 <= t < T) owns m[t] rows (m non-increasing, m[T] = 0 or the rows the state after the last step is written to);
 * BOTH host tables m and row0 therefore hold T + 1 entries and say so in n_m / n_row0 (checked);
 * in every step-packed array ([rows, .]: H1, H2, Gx, G1, G2, AH, AL, dP1, dP2, dAH, dWa, dBa, dCtx) its rows start at row0[t];
 * Hout / dHout rows of step t start at element hout_off[t] / dhout_off[t] with row pitch ld_hout / ld_dhout (the unpacked
 * decoder keeps them sentence-major [S, T, R]); C1, C2 are [T+1][S][R]; k_out (may be NULL) [T][S][R]; Gf, pre [S, 4R].
 * H1 = [h2_{t-1} | h1_{t-1}] (ld ldH1), H2 = [ctx_t | h1_t | h2_{t-1}] (ld ldH2): bf16 when `bf16` (then Wc1, Wc2, Wq, Hout and
 * dP1 / dP2 / dAH are bf16 too), else fp32.  Attention sets: per sentence (off, lens; shared = 0) or per image
 * (rows_map, B, g, Nn; shared = 1: subgc_attn_*_group).  Host arrays (m, row0, hout_off, dhout_off) are read during the call only. */
typedef struct SubgcRecurrence {
    int32_t S;
    int32_t T;
    int32_t R;
    int32_t A;
    int32_t n_alpha;
    int32_t bf16;
    int32_t shared;
    int32_t B;
    int32_t g;
    int32_t Nn;
    int32_t gemm_flags;
    int32_t uv_b16;
    float keep_scale;
    int32_t du_planes;
    int32_t n_m;                 /* entries of m[]    (host array): >= T + 1 -- the forward reads m[T], the rows the last step's state goes to */
    int32_t n_row0;              /* entries of row0[] (host array): >= T + 1 -- the forward reads row0[T], where those rows start              */
    const int32_t* m;
    const int64_t* row0;
    const int64_t* hout_off;
    const int64_t* dhout_off;
    int64_t ld_hout;
    int64_t ld_dhout;
    void* H1;
    int64_t ldH1;
    void* H2;
    int64_t ldH2;
    void* Hout;
    const void* Wc1;
    int64_t ldW1;
    const void* Wc2;
    int64_t ldW2;
    const void* Wq;
    int64_t ldWq;
    const float* b1i;
    const float* b1h;
    const float* b2i;
    const float* b2h;
    const float* bq;
    float* pre;
    const float* Gx;
    const float* Gf;
    float* C1;
    float* C2;
    float* G1;
    float* G2;
    float* AH;
    float* AL;
    const uint8_t* k_out;
    float* QP;
    size_t qp_bytes;
    const void* u;
    const void* v;
    const float* w_a;
    const float* b_a;
    const int32_t* off;
    const int32_t* lens;
    const int32_t* rows_map;
    const float* dHout;
    void* dP1;
    void* dP2;
    void* dAH;
    float* du;
    int64_t du_plane_stride;
    float* dv;
    float* dWa;
    float* dBa;
    float* dCtx;
    float* dE;                   /* per-sentence sets only, may be NULL: deferred d(u) -- step t files d(e) rows at dE + row0[t] * n_alpha, du is not touched */
    float* PA;
    size_t pa_bytes;
    float* PB;
    size_t pb_bytes;
    float* PC;
    size_t pc_bytes;
    float* dC1_in;
    float* dC1_out;
    float* dC2_in;
    float* dC2_out;
} SubgcRecurrence;
int subgc_recurrence_sizeof(void);      /* sizeof(SubgcRecurrence): bindings that mirror the struct check their layout against it */
/* forward: steps 0 .. T-1 in order.  workspace / ws_bytes: the split-K scratch of the cell products (as subgc_lstm_fwd_gemm). */
int subgc_recurrence_fwd(const SubgcRecurrence* a, void* workspace, size_t ws_bytes, void* stream);
/* backward: steps T-1 .. 0.  dC*_in: zeroed [S, R] (the cell-state gradient entering the last step), dC*_out: scratch [S, R]; the two
 * swap roles every step.  dv != NULL: d(v) accumulates per step (dCtx unused); dv == NULL: this step's d(ctx) rows are kept in dCtx
 * for one subgc_attn_dv_accum* after the loop (always so for shared sets). */
int subgc_recurrence_bwd(const SubgcRecurrence* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SUBGC_HIP_H */
