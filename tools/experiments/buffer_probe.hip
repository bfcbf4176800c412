// Semantics of raw buffer loads / stores the macro-element kernel relies on (gfx950): 8-byte accesses
// at 4-byte aligned offsets, 16-byte accesses at 8-byte aligned offsets, out-of-range offsets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float *src, float *dst, unsigned bytes, unsigned oob, float *out)
{
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, bytes, 0x00020000);
  const unsigned t = threadIdx.x;
  // pairs of floats at offsets 4*(2t+1): 4-byte aligned, not 8-byte aligned
  u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, 4 * (2 * t + 1), 0, 0);
  __builtin_amdgcn_raw_buffer_store_b64(v, rd, 4 * (2 * t + 1), 0, 0);
  // out-of-range pair load / store
  u32x2 z = __builtin_amdgcn_raw_buffer_load_b64(rs, oob, 0, 0);
  out[2 * t] = __builtin_bit_cast(float, z.x);
  out[2 * t + 1] = __builtin_bit_cast(float, z.y);
  u32x2 w = {0x7fc00000u, 0x7fc00000u};
  __builtin_amdgcn_raw_buffer_store_b64(w, rd, oob, 0, 0);
  u32x4 w4 = {0x7fc00000u, 0x7fc00000u, 0x7fc00000u, 0x7fc00000u};
  __builtin_amdgcn_raw_buffer_store_b128(w4, rd, oob, 0, 0);
  u32x4 z4 = __builtin_amdgcn_raw_buffer_load_b128(rs, oob, 0, 0);
  out[128 + t] = __builtin_bit_cast(float, z4.x) + __builtin_bit_cast(float, z4.y) + __builtin_bit_cast(float, z4.z) + __builtin_bit_cast(float, z4.w);
}
int main()
{
  const int n = 256;
  std::vector<float> h(n), d(n, -1.f), o(256, -2.f);
  for (int i = 0; i < n; ++i) h[i] = i;
  float *src, *dst, *out;
  hipMalloc(&src, 4 * n); hipMalloc(&dst, 4 * n); hipMalloc(&out, 4 * 256);
  for (unsigned oob : {0xFFFFFFFFu, 0xFFFFFFF0u, 0x80000000u})
    {
      hipMemcpy(src, h.data(), 4 * n, hipMemcpyHostToDevice);
      hipMemcpy(dst, d.data(), 4 * n, hipMemcpyHostToDevice);
      probe<<<1, 64>>>(src, dst, 4 * n, oob, out);
      std::vector<float> r(n), q(256);
      hipMemcpy(r.data(), dst, 4 * n, hipMemcpyDeviceToHost);
      hipMemcpy(q.data(), out, 4 * 256, hipMemcpyDeviceToHost);
      int bad = 0, nanw = 0, oobnz = 0;
      for (int i = 1; i < 129; ++i) bad += r[i] != h[i];
      for (int i = 0; i < n; ++i) nanw += r[i] != r[i];
      for (int i = 0; i < 192; ++i) oobnz += q[i] != 0.f;
      printf("oob=%08x: unaligned pair copy mismatches %d, NaNs written by out-of-range stores %d (dst[0]=%g dst[129]=%g), nonzero out-of-range loads %d\n", oob, bad, nanw, r[0], r[129], oobnz);
    }
  return 0;
}
