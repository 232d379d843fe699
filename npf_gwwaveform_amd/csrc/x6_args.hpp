// What the two program interpreters share (csrc/x6_kernel.hip: fp32 results from three-term bf16 splits; csrc/b16_kernel.hip: the
// bf16 compute mode, one bf16 term): the launch argument block, its validation, and the small device helpers.
#pragma once
#include "npf_common.hpp"

namespace npf {

typedef __bf16 xp_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned xp_u32x4 __attribute__((ext_vector_type(4)));

struct XpArgs {
  npf_x6_op_t op[NPF_X6_MAX_OPS];
  const char* mm_img[NPF_X6_MAX_OPS];  // the multiplies of the program in order (what the slab stream walks)
  int64_t mm_stride[NPF_X6_MAX_OPS];
  const float* mm_bias[NPF_X6_MAX_OPS];  // their bias rows (or null) and per-task strides in floats
  int64_t mm_bias_stride[NPF_X6_MAX_OPS];
  const float* out_w;
  const float* out_b;
  float* out_rows;
  int32_t n_ops, n_mm;
  int32_t total_tiles, tiles_per_task;
  int32_t wgs_per_task;  // 0: tiles dealt flat
  int32_t xcd_remap;     // the workgroups of a task on one XCD (grid a multiple of 8)
  int32_t pts_per_task;  // valid points per task (row-major operands; PT32 operands are padded to whole tiles)
};

__device__ __forceinline__ void xp_dma16(const void* src, void* lds_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_uniform, 16, 0, 0);
}

__device__ __forceinline__ unsigned xp_cvt_pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// sum / max over the four lanes (g = 0..3) that share a point
__device__ __forceinline__ float xp_sum4(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float xp_max4(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

// The program's op table in LDS.  The descriptors are kernel arguments: every `a.op[l].field` is a scalar load whose first touch
// of a descriptor misses the scalar cache and costs a memory round trip at the top of EVERY op (tools/stamp_probe_b16.py: 2200
// cycles per op for the fields alone, more for the pointers the input side tests) -- and a scalar load cannot be issued ahead.
// So the workgroup copies the table to LDS once (vector loads, one round trip; xp_stage_ops) and an op is fetched by broadcast
// ds_read_b64 + readfirstlane (xp_lds_op), as csrc/chain_kernel.hip does for its bf16 instance.
constexpr int kXpOpDwords = sizeof(npf_x6_op_t) / 4;
constexpr int kXpMmDwords = 4 * NPF_X6_MAX_OPS * 2;                           // mm_img, mm_stride, mm_bias, mm_bias_stride
constexpr int kXpTableBytes = (NPF_X6_MAX_OPS * kXpOpDwords + kXpMmDwords) * 4;
static_assert(sizeof(npf_x6_op_t) == 168 && kXpOpDwords == 42, "npf_x6_op_t layout");

// (call once per workgroup, then __syncthreads() before the first xp_lds_op)
__device__ __forceinline__ void xp_stage_ops(const XpArgs& a, unsigned* table, int tid, int n_threads) {
  for (int t = tid; t < a.n_ops * kXpOpDwords; t += n_threads) table[t] = ((const unsigned*)a.op)[t];
  unsigned* mm = table + NPF_X6_MAX_OPS * kXpOpDwords;
  for (int t = tid; t < 2 * NPF_X6_MAX_OPS; t += n_threads) {
    mm[t] = ((const unsigned*)a.mm_img)[t];
    mm[2 * NPF_X6_MAX_OPS + t] = ((const unsigned*)a.mm_stride)[t];
  }
}

// a 64-bit entry of the table as a wave-uniform value (``byte_addr``: its LDS address)
__device__ __forceinline__ unsigned long long xp_lds_u64(unsigned byte_addr) {
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  u32x2_t r;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(byte_addr));
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)r[1]) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)r[0]);
}

__device__ __forceinline__ npf_x6_op_t xp_lds_op(unsigned table_addr, int l) {
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const unsigned a = table_addr + (unsigned)l * (kXpOpDwords * 4);
  u32x2_t r[21];
  asm volatile(
      "ds_read_b64 %0, %21\n\tds_read_b64 %1, %21 offset:8\n\tds_read_b64 %2, %21 offset:16\n\tds_read_b64 %3, %21 offset:24\n\t"
      "ds_read_b64 %4, %21 offset:32\n\tds_read_b64 %5, %21 offset:40\n\tds_read_b64 %6, %21 offset:48\n\tds_read_b64 %7, %21 offset:56\n\t"
      "ds_read_b64 %8, %21 offset:64\n\tds_read_b64 %9, %21 offset:72\n\tds_read_b64 %10, %21 offset:80\n\tds_read_b64 %11, %21 offset:88\n\t"
      "ds_read_b64 %12, %21 offset:96\n\tds_read_b64 %13, %21 offset:104\n\tds_read_b64 %14, %21 offset:112\n\tds_read_b64 %15, %21 offset:120\n\t"
      "ds_read_b64 %16, %21 offset:128\n\tds_read_b64 %17, %21 offset:136\n\tds_read_b64 %18, %21 offset:144\n\tds_read_b64 %19, %21 offset:152\n\t"
      "ds_read_b64 %20, %21 offset:160\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]),
        "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15]), "=&v"(r[16]), "=&v"(r[17]),
        "=&v"(r[18]), "=&v"(r[19]), "=&v"(r[20])
      : "v"(a));
  unsigned long long q[21];
#pragma unroll
  for (int k = 0; k < 21; ++k)
    q[k] = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)r[k][1]) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)r[k][0]);
  // (a generic pointer rebuilt from integers makes every access through it a flat_load / flat_store -- slower to issue, counted
  // on vmcnt AND lgkmcnt; built as global pointers, the accesses stay global_*)
  typedef const __attribute__((address_space(1))) void* gc_t;
  typedef __attribute__((address_space(1))) void* gw_t;
  static_assert(offsetof(npf_x6_op_t, w_img) == 80 && offsetof(npf_x6_op_t, store_bits) == 128 &&
                offsetof(npf_x6_op_t, in_n) == 136, "npf_x6_op_t layout");
  npf_x6_op_t o;
  o.in_pt = (const float*)(gc_t)q[0];
  o.in_rows = (const float*)(gc_t)q[1];
  o.in_w = (const float*)(gc_t)q[2];
  o.in_b = (const float*)(gc_t)q[3];
  o.pre_add = (const float*)(gc_t)q[4];
  o.mask = (const float*)(gc_t)q[5];
  o.mask_bits = (const unsigned long long*)(gc_t)q[6];
  o.sbwd_p = (const float*)(gc_t)q[7];
  o.store_in = (float*)(gw_t)q[8];
  o.store_in_bits = (unsigned long long*)(gw_t)q[9];
  o.w_img = (const void*)(gc_t)q[10];
  o.w_task_stride = (int64_t)q[11];
  o.bias = (const float*)(gc_t)q[12];
  o.bias_task_stride = (int64_t)q[13];
  o.addend = (const float*)(gc_t)q[14];
  o.store_out = (float*)(gw_t)q[15];
  o.store_bits = (unsigned long long*)(gw_t)q[16];
  o.in_n = (int32_t)(unsigned)q[17];
  o.in_relu = (int32_t)(q[17] >> 32);
  o.relu = (int32_t)(unsigned)q[18];
  o.softmax_n = (int32_t)(q[18] >> 32);
  o.softmax_scale = __uint_as_float((unsigned)q[19]);
  o.sbwd_scale = __uint_as_float((unsigned)(q[19] >> 32));
  o.reserved[0] = (int32_t)(unsigned)q[20];
  o.reserved[1] = (int32_t)(q[20] >> 32);
  return o;
}

// Validates a program and fills ``a`` (everything but the grid geometry: wgs_per_task, xcd_remap).  ``flag_mask``: the op flags
// (reserved[0]) this interpreter knows; ``bits_max_width``: the widest program that may carry ReLU-bit / softmax operands.
inline int xp_fill_args(const npf_x6_op_t* ops, int32_t n_ops, const float* out_w, const float* out_b, float* out_rows,
                        int32_t n_tasks, int32_t tiles_per_task, int32_t pts_per_task, int32_t per_task, int32_t width,
                        int32_t flag_mask, int32_t bits_max_width, XpArgs& a) {
  if (!ops || n_ops <= 0 || n_ops > NPF_X6_MAX_OPS || n_tasks <= 0 || tiles_per_task <= 0) return NPF_EINVAL;
  if (pts_per_task <= 0 || pts_per_task > tiles_per_task * 32 || pts_per_task <= (tiles_per_task - 1) * 32) return NPF_EINVAL;
  if ((out_rows != nullptr) != (out_w != nullptr) || (out_b != nullptr && out_rows == nullptr)) return NPF_EINVAL;
  if ((((uintptr_t)out_w) | ((uintptr_t)out_rows)) & 15) return NPF_EINVAL;
  if (((uintptr_t)out_b) & 3) return NPF_EINVAL;
  a.n_mm = 0;
  bool have_cur = false;
  for (int l = 0; l < n_ops; ++l) {
    const npf_x6_op_t& o = ops[l];
    if ((o.in_pt != nullptr) && (o.in_rows != nullptr)) return NPF_EINVAL;
    if ((o.in_rows != nullptr) != (o.in_w != nullptr) || (o.in_b != nullptr && o.in_rows == nullptr)) return NPF_EINVAL;
    if (o.in_rows != nullptr && (o.in_n <= 0 || o.in_n > width || (o.in_n & 15))) return NPF_EINVAL;
    if (o.in_pt != nullptr || o.in_rows != nullptr) have_cur = true;
    if (!have_cur) return NPF_EINVAL;  // (the first op must bring an input)
    if (o.reserved[0] & ~flag_mask) return NPF_EINVAL;
    if (((o.reserved[0] & NPF_X6_IN_RM) && !o.in_pt) || ((o.reserved[0] & NPF_X6_ADD_RM) && !o.addend)) return NPF_EINVAL;
    if ((((uintptr_t)o.in_pt) | ((uintptr_t)o.in_rows) | ((uintptr_t)o.in_w) | ((uintptr_t)o.in_b) | ((uintptr_t)o.pre_add) |
         ((uintptr_t)o.mask) | ((uintptr_t)o.sbwd_p) | ((uintptr_t)o.store_in) | ((uintptr_t)o.w_img) | ((uintptr_t)o.addend) |
         ((uintptr_t)o.store_out)) & 15)
      return NPF_EINVAL;
    if ((((uintptr_t)o.mask_bits) | ((uintptr_t)o.store_in_bits) | ((uintptr_t)o.store_bits)) & 7) return NPF_EINVAL;
    // (a lane's ReLU bits are one 64-bit word: 16 blocks of 4 features; wider programs are inference programs)
    if (width > bits_max_width && (o.mask_bits || o.store_in_bits || o.store_bits || o.sbwd_p || o.softmax_n)) return NPF_EINVAL;
    if (((uintptr_t)o.bias) & 3) return NPF_EINVAL;
    if ((o.w_task_stride & 15) || o.w_task_stride < 0 || o.bias_task_stride < 0) return NPF_EINVAL;
    if ((o.w_task_stride != 0 || o.bias_task_stride != 0) && !per_task) return NPF_EINVAL;
    if (o.softmax_n < 0 || o.softmax_n > width) return NPF_EINVAL;
    if (o.w_img == nullptr && (o.bias || o.addend || o.store_out || o.store_bits || o.relu || o.softmax_n)) return NPF_EINVAL;
    a.op[l] = o;
    if (o.w_img != nullptr) {
      a.mm_img[a.n_mm] = (const char*)o.w_img;
      a.mm_stride[a.n_mm] = o.w_task_stride;
      a.mm_bias[a.n_mm] = o.bias;
      a.mm_bias_stride[a.n_mm] = o.bias_task_stride;
      ++a.n_mm;
    }
  }
  for (int l = n_ops; l < NPF_X6_MAX_OPS; ++l) a.op[l] = ops[0];
  for (int j = a.n_mm; j < NPF_X6_MAX_OPS; ++j) {
    a.mm_img[j] = a.n_mm ? a.mm_img[0] : nullptr;
    a.mm_stride[j] = 0;
    a.mm_bias[j] = nullptr;
    a.mm_bias_stride[j] = 0;
  }
  a.out_w = out_w;
  a.out_b = out_b;
  a.out_rows = out_rows;
  a.n_ops = n_ops;
  a.total_tiles = n_tasks * tiles_per_task;
  a.tiles_per_task = tiles_per_task;
  a.pts_per_task = pts_per_task;
  a.wgs_per_task = 0;
  a.xcd_remap = 0;
  return NPF_OK;
}

}  // namespace npf
