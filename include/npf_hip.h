/*
 * npf_hip.h -- C ABI of the MI355X (gfx950) neural-process hot path.
 *
 * The reference (MarinerQ/npf_GWwaveform) is 100 % Python and has no FFI; the calls below
 * are what a maintainer would bind (ctypes) to replace the torch op
 * sequences of these reference functions (paths relative to the reference root):
 *
 *   npf_chain_run        MLP.forward                      npf/architectures/mlp.py:95-109
 *                        MergeFlatInputs.forward          npf/architectures/encoders.py:175-183
 *                        BaseAttender.forward / DotAttender.score
 *                                                         npf/architectures/attention.py:129-164,204-220
 *                        merge_r_z                        npf/neuralproc/base.py:554-575
 *                        (and the autograd backward of each: dgrad chains)
 *   npf_wgrad_run        autograd of nn.Linear weights/biases on the path (mlp.py:84-91) and
 *                        of torch.bmm / einsum wrt keys/values (attention.py:151,212)
 *   npf_gauss_head_fwd   NeuralProcessFamily.decode tail  npf/neuralproc/base.py:350-365 (+ :116),
 *                        pool_and_replicate_middle        npf/neuralproc/helpers.py:21-32,
 *                        sum_log_prob                     npf/losses.py:18-24
 *   npf_gauss_head_bwd   autograd of the above
 *   npf_mc_objective_fwd/bwd  mean / logsumexp / SUMO over the latent samples
 *                                                         npf/losses.py:146,197-200,262-274
 *   npf_mean_agg_fwd/bwd torch.mean(R_cntxt, dim=1)       npf/neuralproc/np.py:95, attnnp.py:181
 *   npf_pack_pt/unpack_pt  layout change at the module boundary (no reference counterpart)
 *   npf_transpose        W -> W^T for the dgrad chains (no reference counterpart)
 *   npf_cast_bf16_weights  bf16 weight images for the bf16 compute mode (no reference counterpart)
 *   npf_prepare_weights    the two above, batched over the layers of a chain (no reference counterpart)
 *   npf_gather_points    CntxtTrgtGetter.select              npf/utils/datasplit.py:246-255
 *   npf_split_heads/npf_merge_heads  MultiheadAttender._make_multiheaded / _concatenate_multiheads
 *                                                         npf/architectures/attention.py:505-527
 *
 * Conventions: plain device pointers + sizes, caller owns all memory, every call is
 * asynchronous on `stream` (a hipStream_t passed as void*), returns 0 on success and a
 * negative NPF_E* code otherwise (never throws, never allocates, never synchronises).
 * All tensors fp32.
 *
 * "PT32" layout (the on-device layout between kernels): points are grouped in tiles of 32
 * (each task padded to whole tiles); a tile of F features (F % 32 == 0) is stored as
 * [F/4][32 points][4 features], tiles of one task are consecutive, tasks are consecutive:
 *   elem(task, p, f) = (((task*tiles_per_task + p/32) * (F/4) + f/4) * 32 + p%32) * 4 + f%4
 * It is exactly the MFMA 32x32 accumulator layout with the point on the lane, so chain
 * kernels load/store it with fully coalesced 16-byte accesses.
 */
#ifndef NPF_HIP_H
#define NPF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPF_OK 0
#define NPF_EINVAL (-1)   /* bad argument / unsupported size           */
#define NPF_ELAUNCH (-2)  /* hipLaunch failed (hipGetLastError != 0)    */

#define NPF_MAX_OPS 40
#define NPF_MAX_FEATURES 512 /* widest activation a chain keeps in registers, forward and backward (programs that
                                stay <= 256 wide run the 2-workgroups-per-CU variants of the kernel) */

/* ---- chain programs ------------------------------------------------------------- */
enum npf_opcode {
  NPF_OP_END = 0,
  NPF_OP_LOAD_PT = 1,      /* cur <- PT32 tensor p0 (i0 = features F)                          */
  NPF_OP_STORE_PT = 2,     /* PT32 tensor p0 <- cur (i0 = F)                                   */
  NPF_OP_LOAD_ROWS = 3,    /* cur <- row-major p0 [task][pt][i0], i0 <= 32, zero padded        */
  NPF_OP_STORE_ROWS = 4,   /* row-major p0 [task][pt][i0] <- cur features < i0                 */
  NPF_OP_LINEAR = 5,       /* cur <- act(W cur + b [+ PT addend p2]); see npf_op_t fields       */
  NPF_OP_SOFTMAX = 6,      /* cur <- softmax over features < i0 of f0*cur.  i1 = 1: also store the
                              row statistics (max of cur, sum of exp(f0*(cur-max))) to
                              p0[task][pt][2]; i1 = 2: use the statistics stored in p0 instead
                              of the row's own (one block of a softmax over > 512 keys)          */
  NPF_OP_ADD_PT = 7,       /* cur <- cur + PT32 tensor p0 (i0 = F), optional relu (i1)          */
  NPF_OP_MASK_POS = 8,     /* cur <- (PT32 p0 > 0) ? cur : 0 (i0 = F)      [relu backward]      */
  NPF_OP_ADD_TASKVEC = 9,  /* cur <- cur + p0[task][i0 features] (row-major), optional relu (i1)*/
  NPF_OP_ROWDOT_PT = 10,   /* acc0 <- sum_f cur[f] * PT32 p0[f] (i0 = F)   [softmax bwd delta]  */
  NPF_OP_SOFTMAX_BWD = 11, /* cur <- f0 * P * (cur - acc0), P = PT32 p0 (i0 = F)                */
  NPF_OP_RELU = 12,        /* cur <- max(cur, 0)                                                */
  NPF_OP_SCALE = 13,       /* cur <- f0 * cur                                                   */
  NPF_OP_STORE_TR = 14,    /* row-major p0 [task][i0 features][i1 >= 32*tiles points] <- cur: a
                              feature-major copy, i.e. the activations as NPF_W_ROWMAJOR per-task
                              weights W[n = feature][k = point] (s0 = i0*i1, i3 = i1)            */
  NPF_OP_LAYERNORM = 15,   /* cur <- (cur - mean) / sqrt(var + f0) * p0[f] + p1[f] over the i0 <= 256
                              features of the point (nn.LayerNorm of TransformerAttender,
                              npf/architectures/attention.py:552-553,583-586); p0 (gamma), p1
                              (beta): 16-byte aligned, zero-padded to a multiple of 32 floats    */
  NPF_OP_STORE_WB = 18,    /* bf16 mode only: p0 [task][i1 >= 32*tiles rows][roundup(i0, 32)] bf16 <- cur, one row per
                              point, features k-permuted like npf_cast_bf16_weights: the activations as a
                              bf16 weight image W[n = point][k = feature] (per-task NPF_W_ROWMAJOR)         */
  NPF_OP_STORE_TRB = 19,   /* bf16 mode only: p0 [task][i0 features][i1 = 32*tiles columns] bf16 <- cur, the
                              transposed image W[n = feature][k = point], points k-permuted in groups of 32  */
  NPF_OP_LOAD_RM = 17,     /* cur <- row-major p0 [task][pt][i0], i0 % 32 == 0, i0 <= 512 (i4 = modulus)  */
  NPF_OP_STORE_MASK = 20,  /* bf16 mode only: PTM tensor p0 <- (cur > 0) as bits (i0 = F <= 256): the ReLU mask of the
                              backward pass -- one 32-bit word per point, lane group g = (feature / 4) % 4 and 128
                              features, [task][tile][ceil(F/128) words][4 g][32 points]; within a word the 16-feature
                              block bb = (feature / 16) % 8, element e = feature % 4 sits at bit 31 - (4 bb + e)        */
  NPF_OP_MASK_BITS = 21,   /* bf16 mode only: cur <- (bit of PTM tensor p0) ? cur : 0 (i0 = F)   [relu backward]      */
  NPF_OP_LAYERNORM_BWD = 16 /* cur = dy on entry; x = PT32 p0 (the forward input, i0 = F), gamma p1:
                              xhat = (x - mean) rstd;  PT32 p2 <- dy * xhat (for dgamma);
                              cur <- rstd (g - mean(g) - xhat mean(g xhat)),  g = dy * gamma      */
};

enum npf_wmode {
  NPF_W_ROWMAJOR = 0, /* W[n][k] at p0 + task*s0 + n*i3 (nn.Linear layout, i3 = row stride)   */
  NPF_W_PT_ROWS = 1,  /* W[n][k] = PT32 tensor p0 (features = k, points = n) of the WG's task:
                         the weights of q.k^T are the task's keys                              */
  NPF_W_PT_COLS = 2   /* W[n][k] = PT32 tensor p0 (features = n, points = k) of the WG's task:
                         the weights of attn.V are the task's values, transposed              */
};

#define NPF_F_RELU 1u
#define NPF_F_ADD_PT 2u /* add PT32 tensor p2 (same F as the output) before the activation     */
#define NPF_F_MASK_PT 4u /* out = (PT32 tensor p2 > 0) ? out : 0 (fused relu backward); not with ADD_PT */
#define NPF_F_P16 16u     /* bf16 mode only: the op's PT tensor (p0 of LOAD_PT / STORE_PT / ADD_PT / MASK_POS /
                            ROWDOT_PT / SOFTMAX_BWD, p2 of a LINEAR's addend or mask) is a PT16 tensor: bf16 tiles
                            [F/8 rows][32 points][8 features], row 4s+g = features {32s+4g+i}, {32s+16+4g+i}     */
#define NPF_F_MASK_BITS 32u /* bf16 mode only: like MASK_PT with the mask as bits, p2 = PTM tensor (NPF_OP_STORE_MASK)   */
#define NPF_F_ADD_RM 8u  /* like ADD_PT with a row-major addend p2 [task][pt][i1] (i1 % 32 == 0): module-
                            boundary tensors enter without a layout pass (inference paths)            */
#define NPF_F_STORE_IN 64u /* bf16 mode only. LINEAR: PT tensor p3 <- the layer's INPUT (i0 features), exactly what NPF_OP_STORE_PT in
                              front of the layer would store; the pipelined layers spread the stores over their stages
                              instead of bursting them between two layers                                       */
#define NPF_F_STORE_BITS 256u /* bf16 mode only. LINEAR with NPF_F_RELU and no addend / mask: PTM tensor p2 <- (output > 0) as
                                 bits, exactly what NPF_OP_STORE_MASK behind the layer would store (i1 <= 256)             */
#define NPF_F_STORE_P16 128u /* bf16 mode only, with NPF_F_STORE_IN: p3 is a PT16 tensor (see NPF_F_P16)                */

typedef struct npf_op {
  int32_t op;        /* npf_opcode                                                              */
  int32_t i0;        /* LINEAR: K (valid inputs)        others: see opcode                      */
  int32_t i1;        /* LINEAR: N (valid outputs)                                               */
  int32_t i2;        /* LINEAR: npf_wmode                                                       */
  int32_t i3;        /* LINEAR: row stride of W in floats (mode 0); tiles of the PT weight
                        tensor per task (modes 1, 2)                                            */
  uint32_t flags;    /* NPF_F_*                                                                 */
  float f0;          /* SOFTMAX / SOFTMAX_BWD / SCALE: scale                                    */
  int32_t i4;        /* *_PT ops and LINEAR addend: task modulus (0 = none): the tensor's task
                        index is task % i4 (broadcast of X_trgt over n_z samples)               */
  const void *p0;    /* LINEAR: W                        others: tensor                         */
  const void *p1;    /* LINEAR: bias or NULL                                                    */
  const void *p2;    /* LINEAR: PT32 addend or NULL                                             */
  int64_t s0;        /* LINEAR: per-task stride of W in floats (0 = shared)                     */
  int64_t s1;        /* LINEAR: per-task stride of bias in floats (0 = shared)                  */
  void *p3;          /* LINEAR with NPF_F_STORE_IN: destination of the input store              */
} npf_op_t;

typedef struct npf_program {
  int32_t n_ops;
  int32_t n_tasks;        /* tasks in the batch                                                 */
  int32_t pts_per_task;   /* valid points per task                                              */
  int32_t tiles_per_task; /* ceil(pts_per_task / 32)                                            */
  int32_t wg_per_task;    /* 1: every workgroup (4 tiles) stays inside one task (required by
                             per-task weights, modes 1/2); 0: tiles are dealt flat               */
  int32_t reserved[3];
  npf_op_t ops[NPF_MAX_OPS];
} npf_program_t;

/* Runs a chain program: every wavefront keeps the activations of one tile of 32 points in
 * registers (feature-major MFMA accumulator layout) across all ops; weights stream through
 * LDS.  Replaces the reference functions listed at the top of this file. */
int npf_chain_run(const npf_program_t *prog, void *stream);

/* ---- weight/bias gradients -------------------------------------------------------- */
/* One job: dW[n][k] (+)= sum_p dZ[n][p] * A[k][p] and db[n] (+)= sum_p dZ[n][p], where dZ and
 * A are PT32 tensors with roundup(N,32) / roundup(K,32) features over the same points.
 * per_task == 0: one dW (row-major, row stride ldw) for all points of all tasks, reduced
 *                over the workgroups through `partials` (deterministic, no atomics);
 * per_task == 1: one dW per task, written as a PT32 tensor whose *points* are n and whose
 *                features are k ([task][ceil(N/32) tiles][roundup(K,32)/4][32][4]): this is
 *                the gradient of attention keys / values; db is ignored. */
typedef struct npf_wgrad_job {
  const float *dZ;
  const float *A;
  float *dW;
  float *db;        /* may be NULL */
  int64_t ldw;      /* row stride of dW in floats (per_task == 0) */
  int32_t N, K;
  int32_t per_task;
  int32_t accumulate; /* bit 0: 0 = overwrite dW/db, 1 = add to them; bit 1 (NPF_WGRAD_BF16): round the
                         operands to bf16 at the MFMA input (bf16 compute mode; all jobs of a launch alike) */
  /* A job covers at most 256 x 256 of dW.  Wider layers (up to NPF_MAX_FEATURES = 512 features per side, the
   * reference's MLPs have no limit: npf/architectures/mlp.py:44-93) are run as several jobs on blocks of the
   * operands: dZ / A then point at the block's first feature inside the (wider) tensor and ldz / lda give the
   * features per tile of that tensor (0 = the job's own roundup(N, 32) / roundup(K, 32)); ldo likewise for the
   * PT32 output of a per-task job whose K is a block of a wider tensor. */
  int32_t ldz, lda, ldo;
  int32_t reserved;
} npf_wgrad_job_t;
#define NPF_WGRAD_ACCUMULATE 1
#define NPF_WGRAD_BF16 2
#define NPF_WGRAD_DZ16 4 /* with NPF_WGRAD_BF16: dZ is a PT16 tensor (bf16 tiles, see NPF_F_P16) */
#define NPF_WGRAD_A16 8  /* with NPF_WGRAD_BF16: A is a PT16 tensor */
/* fp32 result on the bf16 matrix pipe (all jobs of a launch alike, not with NPF_WGRAD_BF16): each fp32 operand is split
 * exactly into three bf16 terms and six of the nine cross products are accumulated in fp32 -- the dropped terms are
 * below 2^-26 of a product, i.e. under fp32 rounding; same operands, same result to summation-order noise, 6/16 of
 * the v_mfma_f32_16x16x4_f32 time (csrc/wgrad_kernel.hip, wgrad_x6_kernel). */
#define NPF_WGRAD_F32X6 16
/* with NPF_WGRAD_F32X6: keep wgrad_x6_kernel (every wave splits the fragments it reads) for this launch; default for launches whose
 * jobs are all 256 x 256 is wgrad_h16_kernel (every operand value split once per workgroup).  An A/B switch. */
#define NPF_WGRAD_NO_H16 32

#define NPF_MAX_WGRAD_JOBS 16
/* Runs up to NPF_MAX_WGRAD_JOBS jobs over the same points in ONE launch (+ one reduce). */
int npf_wgrad_run(const npf_wgrad_job_t *jobs, int32_t n_jobs, int32_t n_tasks, int32_t tiles_per_task,
                  float *partials, int64_t partials_bytes, void *stream);
/* Bytes of workspace npf_wgrad_run needs for these jobs (-1 on invalid arguments). */
int64_t npf_wgrad_partials_bytes(const npf_wgrad_job_t *jobs, int32_t n_jobs, int32_t n_tasks,
                                 int32_t tiles_per_task);

/* ---- Gaussian head ------------------------------------------------------------------ */
/* suff: row-major [n_rows][pts][2*dy] raw decoder output.  loc/scale: [n_rows][pts][dy].
 * scale = 0.01 + 0.99 softplus(raw) (base.py:116); homoskedastic != 0 pools scale over the
 * points of each row-task (base.py:356-362).  If Y != NULL (row-major [n_y_rows][pts][dy],
 * row r uses Y[r % n_y_rows]) also writes sum_logp[n_rows] = sum_t sum_dy log N(y|loc,scale)
 * (losses.py:18-24).  loc == scale == NULL: loss-only launch, only sum_logp is written (nothing of size
 * [n_rows][pts][dy] is materialised: the multi-sample objectives below only need sum_logp). */
int npf_gauss_head_fwd(const float *suff, int32_t n_rows, int32_t pts, int32_t dy, int32_t homoskedastic,
                       const float *Y, int32_t n_y_rows, float *loc, float *scale, float *sum_logp,
                       void *stream);
/* d_suff from (d_loc, d_scale, d_sum_logp): any of the three upstream gradients may be NULL.  loc == scale == NULL
 * (after a loss-only forward): they are recomputed from suff; d_loc and d_scale must then be NULL. */
int npf_gauss_head_bwd(const float *suff, const float *loc, const float *scale, int32_t n_rows, int32_t pts,
                       int32_t dy, int32_t homoskedastic, const float *Y, int32_t n_y_rows,
                       const float *d_loc, const float *d_scale, const float *d_sum_logp, float *d_suff,
                       void *stream);

/* ---- Monte-Carlo objectives over the latent samples (npf/losses.py:126-276) ------------ */
/* log_w: row-major [n_z][n_tasks], the log weight of latent sample k for task b: sum_t log p(y_t | z_k), plus
 * log q(z_k | C) - log q(z_k | C, T) when the samples come from q(z | C, T).  out[n_tasks]:
 *   mode 0  mean_k log_w                                  (E_z of ELBOLossLNPF, losses.py:146)
 *   mode 1  logsumexp_k log_w - log n_z                   (NLLLossLNPF, losses.py:197-200)
 *   mode 2  SUMO (SUMOLossLNPF, losses.py:262-274): c_k = logsumexp_{j<=k} log_w_j - log(k+1) (k 0-based),
 *           out = c_{m-1} + sum_{k>=m} inv_weights[k] (c_k - c_{k-1}); inv_weights[n_z] = P(K >= k) of the
 *           number-of-samples distribution (host), m = the smallest number of samples it draws.
 * Running logsumexp per task: no [n_z][n_tasks] temporaries (and, with the loss-only Gaussian head above,
 * nothing of size [n_z][n_tasks][targets][dy]). */
int npf_mc_objective_fwd(const float *log_w, int32_t n_z, int32_t n_tasks, int32_t mode, const float *inv_weights,
                         int32_t m, float *out, void *stream);
/* d_log_w[n_z][n_tasks] from d_out[n_tasks]; workspace: n_z * n_tasks floats (mode 2 only, else may be NULL). */
int npf_mc_objective_bwd(const float *log_w, int32_t n_z, int32_t n_tasks, int32_t mode, const float *inv_weights,
                         int32_t m, const float *d_out, float *d_log_w, float *workspace, void *stream);

/* ---- mean aggregation over the points of a task ------------------------------------- */
/* out[task][F] (row-major) = mean over valid points of PT32 tensor R (np.py:95). */
int npf_mean_agg_fwd(const float *R_pt, int32_t n_tasks, int32_t pts_per_task, int32_t F, float *out,
                     void *stream);
/* dR_pt[task][p][f] = d_out[task][f] / pts_per_task for valid points, 0 for padding. */
int npf_mean_agg_bwd(const float *d_out, int32_t n_tasks, int32_t pts_per_task, int32_t F, float *dR_pt,
                     int32_t accumulate, void *stream);

/* ---- layout ----------------------------------------------------------------------- */
/* rows: row-major [n_tasks][pts_per_task][F_valid]; pt: PT32 with F = roundup(F_valid, 32). */
int npf_pack_pt(const float *rows, int32_t n_tasks, int32_t pts_per_task, int32_t F_valid, float *pt,
                void *stream);
int npf_unpack_pt(const float *pt, int32_t n_tasks, int32_t pts_per_task, int32_t F_valid, float *rows,
                  void *stream);
/* dst[c][r] = src[r][c] for a row-major [rows][cols] matrix. */
int npf_transpose(const float *src, int32_t rows, int32_t cols, float *dst, void *stream);
/* Heads as tasks (MultiheadAttender._make_multiheaded / _concatenate_multiheads,
 * npf/architectures/attention.py:505-527) on PT32 tensors:
 *   split:  dst PT32 [n_heads*n_tasks][pts][F/n_heads]:  dst(h*n_tasks + b, p, f) = src(b, p, h*(F/n_heads) + f)
 *   merge:  the inverse (dst PT32 [n_tasks][pts][F]).  F % n_heads == 0 and (F/n_heads) % 4 == 0. */
int npf_split_heads(const float *src, int32_t n_tasks, int32_t pts_per_task, int32_t F, int32_t n_heads, float *dst,
                    void *stream);
int npf_merge_heads(const float *src, int32_t n_tasks, int32_t pts_per_task, int32_t F, int32_t n_heads, float *dst,
                    void *stream);

/* Multihead scaled-dot attention with 16- or 32-feature heads (D = F / n_heads), straight on the PT32 tensors of the K / Q / V projections
 * (MultiheadAttender.forward between the projections and the concatenation, npf/architectures/attention.py:505-527 with
 * DotAttender :204-220 per head, scale 1 / sqrt(head size)): out(b, q, D h + :) = softmax_k(Q_h K_h^T / sqrt(D)) V_h, X_h = features
 * D h .. D h + D - 1.  q / out: PT32 [n_tasks][n_queries][F], k / v: PT32 [n_tasks][n_keys][F], F = D n_heads (tiles of
 * roundup(F, 32) features), n_keys <= 256 (D = 16) or 128 (D = 32).  lse [n_tasks][n_heads][n_queries] (or NULL at inference): log-sum-exp of the scaled
 * scores, what the backward pass recomputes the probabilities from.  fp32 MFMA, softmax with max subtraction.
 * npf_mha_bwd: d_q / d_k / d_v (PT32, same shapes as q / k / v; points and features outside the valid ranges are not written). */
int npf_mha_fwd(const float *q, const float *k, const float *v, int32_t n_tasks, int32_t n_heads, int32_t n_keys,
                int32_t n_queries, int32_t F, float *out, float *lse, void *stream);
int npf_mha_bwd(const float *q, const float *k, const float *v, const float *out, const float *d_out, const float *lse,
                int32_t n_tasks, int32_t n_heads, int32_t n_keys, int32_t n_queries, int32_t F, float *d_q, float *d_k, float *d_v,
                void *stream);

/* y = LayerNorm(a + b) over the F features of every point (F % 4 == 0, F <= 256), PT32 tensors [n_tasks][pts_per_task][F]:
 * TransformerAttender.forward's layer_norm1(context + queries) (npf/architectures/attention.py:566-575; nn.LayerNorm: biased
 * variance, eps inside the root, gamma / beta [F]).  stats [n_tasks * tiles * 32][2] = (mean, 1 / std) per point, or NULL at inference.
 * npf_add_layernorm_bwd: dx (the gradient of a and of b alike; zero at the padding points) and partials [n_tasks * tiles][2][F] = the
 * tile's sums of dy * xhat and of dy, which the caller adds up over the tiles to dgamma and dbeta. */
int npf_add_layernorm_fwd(const float *a, const float *b, const float *gamma, const float *beta, float eps, int32_t n_tasks,
                          int32_t pts_per_task, int32_t F, float *y, float *stats, void *stream);
int npf_add_layernorm_bwd(const float *a, const float *b, const float *gamma, const float *stats, const float *dy, int32_t n_tasks,
                          int32_t pts_per_task, int32_t F, float *dx, float *partials, void *stream);

/* Context / target selection (CntxtTrgtGetter.select, npf/utils/datasplit.py:246-255: torch.gather along
 * the points of X and y with the same indices): out_x[b][i][:] = x[b][idx[b][i]][:], same for y.
 * idx: int64 [n_tasks][n_sel], every entry in [0, n_points) (checked on the host side). */
int npf_gather_points(const float *x, const float *y, const int64_t *idx, int32_t n_tasks, int32_t n_points,
                      int32_t n_sel, int32_t x_dim, int32_t y_dim, float *out_x, float *out_y, void *stream);

/* bf16 compute mode of the chain kernel (prog->reserved[2] == 1): every LINEAR op takes, instead of fp32
 * weights, the bf16 image this function writes -- dst [rows][roundup(cols, 32)] bf16, columns permuted
 * inside each group of 32 (position 8g+i <- column 4g+i, position 8g+4+i <- column 16+4g+i), zero
 * padded -- with p0 = dst, i3 = roundup(K, 32) / 2 (row stride in floats), NPF_W_ROWMAJOR.  Activations
 * are rounded to bf16 at the MFMA input, accumulation / bias / epilogue / HBM tensors stay fp32.
 * transposed != 0: the image of src^T (rows = columns of src), for the dgrad chains. */
int npf_cast_bf16_weights(const float *src, int32_t n_rows, int32_t n_cols, int32_t ld, int32_t transposed, void *dst,
                          void *stream);

/* Batched weight preparation: up to NPF_MAX_WPREP_JOBS of the two functions above in ONE launch (a chain's
 * dgrad needs W^T -- or, in the bf16 mode, an image -- of every layer; one 5 us launch per layer otherwise).
 * kind: 0 = fp32 transpose (as npf_transpose, src row stride ld), 1 = bf16 image, 2 = bf16 image of src^T;
 * 1 / 2 + 16 t, t = 1..3: the same image of term t - 1 of the exact three-term split of src (x0 = bf16(x), x1 = bf16(x - x0),
 * x2 = bf16(x - x0 - x1)) -- the weights of npf_mlp_x6_run;
 * 5 / 6 = kinds 1 / 2 with every image row zero-padded to 256 inputs (a layer with <= 32 real inputs whose remaining
 * input registers are known to be zero then runs as a 256-input layer on the pipelined path). */
typedef struct npf_wprep_job {
  const float *src; /* row-major [n_rows][n_cols], row stride ld floats */
  void *dst;        /* kind 0: float [n_cols][n_rows]; kind 1 / 2: bf16 image, 16-byte aligned */
  int32_t n_rows, n_cols, ld, kind;
} npf_wprep_job_t;
#define NPF_MAX_WPREP_JOBS 32
int npf_prepare_weights(const npf_wprep_job_t *jobs, int32_t n_jobs, void *stream);

/* Library / device info. */
/* ---- hidden layers of the flat MLPs with their fp32 products on the bf16 matrix pipe ----------------------------------
 * Replaces, for stacks of 256 -> 256 layers, what npf_chain_run does with LINEAR ops (npf/architectures/mlp.py:95-109 forward;
 * its autograd backward): cur <- x; per layer: [cur <- mask > 0 ? cur : 0] [store_in <- cur] cur <- W cur + bias [+ addend] [relu]
 * [store_out <- cur]; y <- cur.  x, y, mask, store_in, store_out: PT32 tensors with 256 features over n_tasks x tiles_per_task
 * tiles.  w_img: the layer's weights as THREE bf16 terms, W = W0 + W1 + W2 exactly to 2^-27 (W0 = bf16(W), W1 = bf16(W - W0),
 * W2 = bf16(W - W0 - W1)), each a 256 x 256 k-permuted image as npf_cast_bf16_weights makes it, stored back to back; the kernel
 * splits the layer input the same way in registers and accumulates six of the nine cross products in fp32 (the dropped ones
 * are below 2^-26 of a product): an fp32 result, at 6/16 of the fp32-MFMA time (csrc/mlp_x6_kernel.hip).
 * Forward pass of a stack: bias, relu, store_out (the saved activations) per layer.  Its dgrad: layers in reverse order with
 * w_img = the images of W^T, mask = the layer's saved output, store_in = the dZ buffer the weight gradient reads. */
#define NPF_X6_MAX_LAYERS 8
typedef struct {
  const void *w_img;   /* [3][256][256] bf16 */
  const float *bias;   /* [256] or NULL */
  const float *mask;   /* PT32 or NULL */
  float *store_in;     /* PT32 or NULL */
  float *store_out;    /* PT32 or NULL */
  const float *addend; /* PT32 or NULL: added before the ReLU (MergeFlatInputs: relu(x1 + resizer(x2)), encoders.py:178-179) */
  /* where the layer's output is positive, as bits: one uint64 per lane and half tile ([tiles][2][64], bit 4 b + e = the
   * lane's block b, element e) -- written by a forward layer (store_bits), read by the dgrad of the same layer (mask_bits)
   * in place of `mask`: 8 bytes per 64 values */
  unsigned long long *store_bits;
  const unsigned long long *mask_bits;
  int32_t relu;
  int32_t reserved;
} npf_x6_layer_t;
int npf_mlp_x6_run(const npf_x6_layer_t *layers, int32_t n_layers, const float *x, float *y, int32_t n_tasks,
                   int32_t tiles_per_task, void *stream);
/* The same with a 256 -> 4 layer on either side of the stack (the decoder's output layer, npf/architectures/mlp.py:109), as rows
 * [n_tasks * tiles_per_task * 32][4] row-major and a [4][256] fp32 matrix, in plain fp32 FMAs:
 *  - in front (x == NULL): x[point][f] = sum_n in_rows[point][n] in_w[n][f] -- the dgrad of that layer ahead of the dgrad of
 *    the stack (in_rows = dOut as the Gaussian head's backward leaves it, in_w = W_out): the 256-wide gradient is never
 *    written to or read from HBM;
 *  - behind (out_rows != NULL): out_rows[point][n] = sum_f out_w[n][f] cur[f] + out_b[n] from the registers the last layer
 *    leaves (forward; y may be NULL when the stack's own output is not needed). */
int npf_mlp_x6_run_rows(const npf_x6_layer_t *layers, int32_t n_layers, const float *x, const float *in_rows, const float *in_w,
                        float *y, const float *out_w, const float *out_b, float *out_rows, int32_t n_tasks,
                        int32_t tiles_per_task, void *stream);

/* ---- x6 programs: whole sides of the model as ONE launch on the bf16 matrix pipe, fp32 results -------------------------
 * The generalisation of npf_mlp_x6_run: a straight-line program of up to NPF_X6_MAX_OPS ops on the register-resident activation
 * `cur` of every point (width F = 128 or 256 features, PT32 tensors with exactly F features), every multiply an fp32 product
 * as six bf16 products of exact three-term splits (see above).  Besides the shared-weight layers of the flat MLPs
 * (npf/architectures/mlp.py:95-109) an op can
 *  - take PER-TASK weights (w_task_stride != 0): the task's keys / values as three-term images (npf_x6_task_images) -- the two
 *    contractions of DotAttender (npf/architectures/attention.py:204-220 scores, :151 attn . values) and of its backward;
 *  - apply the softmax of attention.py:158-164 to its result (softmax_n keys, in registers, wavefront shuffles), or the softmax
 *    backward to its input (sbwd_p);
 *  - form its input from [points][4] rows through a [4][in_n] matrix (+ bias, ReLU): the first layer of the x-encoder and of
 *    the XY-encoder's resizer (Linear(1 -> r), Linear(2 -> 32); mlp.py:96) at no HBM traffic, or the dgrad of a 256 -> 4 layer;
 *  - run without a multiply (w_img == NULL): an elementwise step (mask + store: the end of a dgrad chain).
 * Per op, in this order:
 *   input side   cur <- in_pt                           (a new PT32 input; else cur = the previous op's result)
 *                cur <- [relu](in_w^T rows + in_b)      (in_rows; features >= in_n are zero)
 *                cur += pre_add                         (PT32)
 *                cur <- mask > 0 ? cur : 0              (mask: PT32; mask_bits: one uint64 per lane and half tile, see below)
 *                cur <- sbwd_scale * P * (cur - <cur, P>)   (sbwd_p = P, PT32)
 *                store_in <- cur (PT32);  store_in_bits <- (cur > 0)
 *   multiply     cur <- W cur + bias                    (w_img: [3][F][F] bf16 three-term image, + task * w_task_stride BYTES;
 *                                                        bias [F] or NULL, + task * bias_task_stride floats)
 *   output side  cur += addend (PT32);  relu;  softmax over the first softmax_n features of softmax_scale * cur;
 *                store_out <- cur (PT32);  store_bits <- (cur > 0)
 * Bit tensors: [n_tasks * tiles_per_task][2][64] uint64, bit 4 b + e of lane (p, g) = feature 16 b + 4 g + e of point p of
 * the half tile.  Rows tensors: [n_tasks * tiles_per_task * 32][4] (the points of a task fill whole tiles). */
#define NPF_X6_MAX_OPS 12
typedef struct npf_x6_op {
  const float *in_pt;
  const float *in_rows;
  const float *in_w;
  const float *in_b;
  const float *pre_add;
  const float *mask;
  const unsigned long long *mask_bits;
  const float *sbwd_p;
  float *store_in;
  unsigned long long *store_in_bits;
  const void *w_img;
  int64_t w_task_stride;
  const float *bias;
  int64_t bias_task_stride;
  const float *addend;
  float *store_out;
  unsigned long long *store_bits;
  int32_t in_n;        /* outputs of the rows prologue (a multiple of 16, <= F) */
  int32_t in_relu;
  int32_t relu;
  int32_t softmax_n;   /* 0 = no softmax */
  float softmax_scale;
  float sbwd_scale;
  int32_t reserved[2]; /* [0]: op flags (NPF_X6_IN_RM ...); [1]: 0 */
} npf_x6_op_t;
/* op flags (reserved[0]): in_pt / addend are ROW-MAJOR [n_tasks][pts_per_task][F] tensors -- what the reference's decode(X_trgt_enc,
 * R_trgt) is handed (npf/neuralproc/base.py:327) -- read without a layout pass (inference inputs: no gradient flows into them) */
#define NPF_X6_IN_RM 1
#define NPF_X6_ADD_RM 2
/* npf_b16_run only: store_in / store_out take the fp32 value as a PT32 tensor (default there: bf16(value) as a PT16 tensor) */
#define NPF_X6_STORE_IN_F32 4
#define NPF_X6_STORE_OUT_F32 8
/* 256-feature programs, the library's choice of kernel instance (npf_x6_run_ex variant 0): 1 = 16 points per wave, four waves per
 * workgroup, two workgroups per CU; 2 = 32 points per wave (one wave per SIMD, every weight fragment feeds twice the matrix
 * instructions); 3 = 16 points per wave, EIGHT waves per workgroup sharing one slab ring (a CU streams every slab once) */
#define NPF_X6_DEFAULT_VARIANT 3
/* out_rows != NULL: a F -> 4 layer behind the program, out_rows[point][n] = sum_f out_w[n][f] cur[f] + out_b[n] (the decoder's
 * output layer, mlp.py:109).  per_task != 0: every workgroup stays inside one task (required by per-task weights / biases).
 * width: 128 or 256. */
/* npf_x6_run_ex: the same with pts_per_task valid points per task (<= 32 tiles_per_task; row-major operands are indexed with it),
 * width 128, 256 or 512 (512: no ReLU-bit operands -- inference programs, e.g. the r = 512 decoder of base.py:327-367 from
 * row-major inputs), and variant = 0 (library's choice) | 1 | 2 | 3: width 256 see NPF_X6_DEFAULT_VARIANT above; width 512: 1 = a wave
 * holds all 512 features of its 16 points (one wave per SIMD), 0 / 2 = the contraction split over pairs of waves through LDS
 * (two waves per SIMD; programs of plain layers -- inputs, multiply, bias / addend / ReLU, stores -- only; others take 1). */
int npf_x6_run_ex(const npf_x6_op_t *ops, int32_t n_ops, const float *out_w, const float *out_b, float *out_rows,
                  int32_t n_tasks, int32_t tiles_per_task, int32_t pts_per_task, int32_t per_task, int32_t width,
                  int32_t variant, void *stream);
int npf_x6_run(const npf_x6_op_t *ops, int32_t n_ops, const float *out_w, const float *out_b, float *out_rows,
               int32_t n_tasks, int32_t tiles_per_task, int32_t per_task, int32_t width, void *stream);
/* The three-term images of a PT32 tensor src [n_tasks][tiles_per_task][F/4][32][4] taken as per-task weights, F = width:
 *   row_img [n_tasks][3][F][F] bf16: W[n = point][k = feature]   (scores = K q, dP = V dO), points >= pts are zero rows;
 *   tr_img  [n_tasks][3][F][F] bf16: W[n = feature][k = point]   (attn . V, dq = K^T dS), points >= pts are zero columns;
 * both k-permuted like npf_cast_bf16_weights.  pts <= F points per task.  Either destination may be NULL. */
int npf_x6_task_images(const float *src, int32_t n_tasks, int32_t pts, int32_t width, void *row_img, void *tr_img, void *stream);

/* ---- b16 programs: the same programs in the bf16 compute mode (BASELINE config 3) ------------------------------------------
 * npf_x6_op_t programs of width 128 or 256 with ONE bf16 product per multiply, fp32 accumulation: the input of every multiply is
 * rounded to bf16 (nearest even), w_img is a ONE-term image [F][F] bf16 (npf_prepare_weights kinds 1 / 2; npf_b16_task_images for
 * the task's keys / values), bias / addend / ReLU / softmax in fp32 -- the arithmetic of the bf16 chain instance
 * (npf/architectures/mlp.py:95-109, attention.py:129-164 with the rounding points DESIGN.md 4 lists).  Differences to npf_x6_run:
 *   store_in / store_out  <- bf16(cur) as a PT16 tensor [tiles][F/8][32][8] (see NPF_F_P16) -- what only the weight-gradient launch
 *                            reads -- unless NPF_X6_STORE_IN_F32 / NPF_X6_STORE_OUT_F32 asks for the fp32 value as a PT32 tensor;
 *   sbwd_p                =  a PT16 tensor (the saved probabilities are bf16 values);
 *   in_rows prologue / the F -> 4 layer behind the program: fp32 FMAs; the caller hands bf16-ROUNDED rows and matrices, the
 *                            F -> 4 layer rounds its input itself;
 *   mask (PT32), row-major operands: not available.
 * Whole tiles only (pts_per_task = 32 tiles_per_task).  variant: 0 = the library's choice (1), 1 = eight waves per workgroup on one
 * ring of three 64-row slabs (one workgroup per CU), 2 = four waves and three 32-row slabs (two workgroups per CU). */
int npf_b16_run(const npf_x6_op_t *ops, int32_t n_ops, const float *out_w, const float *out_b, float *out_rows, int32_t n_tasks,
                int32_t tiles_per_task, int32_t per_task, int32_t width, int32_t variant, void *stream);
/* npf_x6_task_images with the rounded value alone: row_img / tr_img [n_tasks][F][F] bf16. */
int npf_b16_task_images(const float *src, int32_t n_tasks, int32_t pts, int32_t width, void *row_img, void *tr_img, void *stream);

int npf_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NPF_HIP_H */
