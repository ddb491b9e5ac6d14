// Shared device/host helpers for the gfx950 U-Net kernels.
//
// Storage types: activations live in HBM as channels-last ("pixel-major") tensors,
// [n][z][y][x][pitch] with the used channels at [coff, coff+c).  Element type is
// bf16 (performance mode) or fp32 (parity mode).  One-channel tensors (image,
// logits and the whole top of the decoder) are always fp32 and therefore
// identical to PyTorch's contiguous NCHW/NCDHW layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MI_NSLOT 8  // partial-sum slots per (image, channel) for instance-norm statistics

typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

// ---- element traits ---------------------------------------------------------
struct BF16 {
    typedef unsigned short elem;
    static constexpr int ESZ = 2;
    static constexpr int VEC = 8;  // elements per 16-byte vector
    static constexpr int DT = 1;
};
struct F32 {
    typedef float elem;
    static constexpr int ESZ = 4;
    static constexpr int VEC = 4;
    static constexpr int DT = 0;
};

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;  // round-to-nearest-even, NaN preserving (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(unsigned short, b);
}

template <class TR>
__device__ __forceinline__ void unpack_vec(const uint4& u, float* v);
template <>
__device__ __forceinline__ void unpack_vec<BF16>(const uint4& u, float* v) {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <>
__device__ __forceinline__ void unpack_vec<F32>(const uint4& u, float* v) {
    v[0] = __uint_as_float(u.x); v[1] = __uint_as_float(u.y);
    v[2] = __uint_as_float(u.z); v[3] = __uint_as_float(u.w);
}
template <class TR>
__device__ __forceinline__ uint4 pack_vec(const float* v);
template <>
__device__ __forceinline__ uint4 pack_vec<BF16>(const float* v) {
    uint4 u;
    u.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
    u.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
    u.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
    u.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
    return u;
}
template <>
__device__ __forceinline__ uint4 pack_vec<F32>(const float* v) {
    uint4 u;
    u.x = __float_as_uint(v[0]); u.y = __float_as_uint(v[1]);
    u.z = __float_as_uint(v[2]); u.w = __float_as_uint(v[3]);
    return u;
}
template <class TR>
__device__ __forceinline__ void load_vec(const void* p, float* v) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    unpack_vec<TR>(u, v);
}
template <class TR>
__device__ __forceinline__ void store_vec(void* p, const float* v) {
    *reinterpret_cast<uint4*>(p) = pack_vec<TR>(v);
}

// CNT (4 or 8) consecutive channels <-> floats
template <class TR, int CNT>
__device__ __forceinline__ void load_n(const char* p, float* v) {
    if (TR::ESZ == 2) {
        if (CNT == 8) {
            uint4 u = *reinterpret_cast<const uint4*>(p);
            unpack_vec<BF16>(u, v);
        } else {
            uint2 u = *reinterpret_cast<const uint2*>(p);
            v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
            v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int i = 0; i < CNT; i += 4) {
            uint4 u = *reinterpret_cast<const uint4*>(p + i * 4);
            v[i] = __uint_as_float(u.x); v[i + 1] = __uint_as_float(u.y);
            v[i + 2] = __uint_as_float(u.z); v[i + 3] = __uint_as_float(u.w);
        }
    }
}
template <class TR, int CNT>
__device__ __forceinline__ void store_n(char* p, const float* v) {
    if (TR::ESZ == 2) {
        if (CNT == 8) {
            *reinterpret_cast<uint4*>(p) = pack_vec<BF16>(v);
        } else {
            uint2 u;
            u.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            u.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            *reinterpret_cast<uint2*>(p) = u;
        }
    } else {
#pragma unroll
        for (int i = 0; i < CNT; i += 4) {
            uint4 u;
            u.x = __float_as_uint(v[i]); u.y = __float_as_uint(v[i + 1]);
            u.z = __float_as_uint(v[i + 2]); u.w = __float_as_uint(v[i + 3]);
            *reinterpret_cast<uint4*>(p + i * 4) = u;
        }
    }
}

// sum over the 64 lanes of a wave
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- counter-based dropout RNG ----------------------------------------------
// Stateless, so backward recomputes the very same mask instead of storing it.  The elements of one image
// are taken in groups of four consecutive indices; a group draws 64 hash bits, 16 per element, from its
// group number under three keys derived from (seed, layer, image); an element is dropped when its 16 bits
// are below the threshold (P(drop) = thresh16 / 65536).  One integer multiply per element: these kernels
// stream at a few VALU slots per byte, and a full hash per element costs more than the memory traffic.
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
struct DropKeys { unsigned k0, k1, k2, cut; };   // cut = thresh16 << 16
__device__ __forceinline__ DropKeys dropout_keys(unsigned long long seed, unsigned layer, unsigned img, unsigned thresh16) {
    unsigned a = mix32((unsigned)seed + 0x9e3779b9u);
    a = mix32(a ^ (unsigned)(seed >> 32));
    a = mix32(a ^ layer);
    a = mix32(a ^ img);
    DropKeys k;
    k.k0 = a; k.k1 = mix32(a ^ 0x5bd1e995u); k.k2 = mix32(a ^ 0xc2b2ae35u);
    k.cut = thresh16 << 16;
    return k;
}
// keep[j] for the elements 4*g + j, j = 0..3, of the image
__device__ __forceinline__ void dropout_keep4(const DropKeys& k, unsigned g, bool* keep) {
    unsigned y = g ^ k.k0;
    y ^= y >> 16; y *= 0x7feb352du;
    unsigned a = y + k.k1, b = y + k.k2;
    a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    b ^= b >> 15; b *= 0x735a2d97u; b ^= b >> 15;
    keep[0] = (a << 16) >= k.cut; keep[1] = a >= k.cut;
    keep[2] = (b << 16) >= k.cut; keep[3] = b >= k.cut;
}
__device__ __forceinline__ bool dropout_keep1(const DropKeys& k, unsigned idx) {
    bool keep[4];
    dropout_keep4(k, idx >> 2, keep);
    const unsigned j = idx & 3u;
    return j == 0 ? keep[0] : j == 1 ? keep[1] : j == 2 ? keep[2] : keep[3];
}

// ---- host side ----------------------------------------------------------------
#include <string>
void mi_set_error(const std::string& s);
// mean / rstd of a one-channel image from its fp64 statistics slots ([nb][NSLOT][2])
__device__ __forceinline__ void c1_moments(const double* stats, int img, int npix, float eps, float& mean, float& rstd) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int sl = 0; sl < MI_NSLOT; ++sl) {
        s1 += stats[((size_t)img * MI_NSLOT + sl) * 2];
        s2 += stats[((size_t)img * MI_NSLOT + sl) * 2 + 1];
    }
    const double m = s1 / (double)npix;
    double var = s2 / (double)npix - m * m;
    var = var > 0.0 ? var : 0.0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}


// hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) once per (kernel, device) and size
// instead of on every launch: the call takes a runtime lock and is a measurable share of the host time of a step
// whose kernels are short.
#include <map>
#include <mutex>
inline hipError_t mi_func_set_lds(const void* kern, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    int& cur = granted[std::make_pair(kern, dev)];
    if (bytes <= cur) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) cur = bytes;
    return e;
}
#define MI_CHECK(cond, msg)                                   \
    do {                                                      \
        if (!(cond)) {                                        \
            mi_set_error(std::string(__func__) + ": " + msg); \
            return -1;                                        \
        }                                                     \
    } while (0)
#define MI_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            mi_set_error(std::string(__func__) + ": " #call ": " + hipGetErrorString(e_)); \
            return -2;                                                                   \
        }                                                                                \
    } while (0)

// "This kernel has started": one relaxed system-scope store by the first workgroup.  A side stream waiting on the word
// (hipStreamWaitValue32) is released without any packet on this kernel's own stream; what it then reads was written by
// kernels that completed, caches released, before this one could start.
__device__ __forceinline__ void signal_start(unsigned* flag, unsigned val) {
    if (flag && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
        __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
