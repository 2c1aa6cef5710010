// tests/emu/isa_interp.cc -- TEST INFRASTRUCTURE.  An interpreter for the gfx950 (CDNA4) machine code hipcc makes of the library's kernels,
// plugged into the tests' emulator: with MGPU_EMU_ISA=<file.s>[:<file.s>...] (`hipcc -S --cuda-device-only` dumps) a launch whose kernel
// is found in one of the files executes THAT INSTRUCTION STREAM -- wave by wave, 64 lanes under an EXEC mask, SGPRs / VGPRs / VCC / SCC,
// LDS, scratch, flat / global memory = the host's -- instead of the host-compiled C++ of the same kernel.  Everything around the launch
// (scene upload, launch parameters, hand-out tables, the other kernels) is the emulator's as before.
//
// What it is for (no GPU at hand):
//   * the code the COMPILER made -- register allocation, the structurizer's exec-mask control flow, OCML's inlined sqrt / division /
//     sincospi sequences, ds_bpermute shuffles, spills -- runs against the oracle, not just the C++ the kernels were written in;
//   * exact dynamic instruction counts by class (VALU / SALU / branch / LDS / VMEM / SMEM, as rocprofv3's SQ_INSTS_* count them) and per
//     instruction (isa_profile_dump), i.e. the quantity the render kernels are bound by, for A/B of kernel changes.
// What it is not: a timing model (no caches, no latencies, no issue rules), a full ISA (the ~230 opcodes the render kernels use; an
// unknown opcode aborts with its source line), or anything the product links.
//
// Semantics follow AMD's "CDNA3 / MI300 Instruction Set Architecture" pseudo-code.  v_rcp_f64 / v_rsq_f64 / v_rcp_f32 return the
// correctly rounded value (the hardware's is within 1 ulp): every use in the kernels is the seed of a Newton iteration or of the
// div_scale / div_fmas / div_fixup sequence, whose results do not depend on the seed's last bits.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace isa {

// ---------------------------------------------------------------------------------------------------------------- operands
enum OKind : uint8_t { K_NONE, K_VGPR, K_SGPR, K_VCC, K_VCC_LO, K_VCC_HI, K_EXEC, K_EXEC_LO, K_EXEC_HI, K_SCC, K_M0, K_IMM, K_OFF, K_LABEL, K_SHARED_BASE, K_PRIVATE_BASE, K_HWREG, K_NULL };
struct Operand {
  OKind kind = K_NONE;
  uint16_t reg = 0;  // first register
  uint8_t n = 1;     // registers
  bool neg = false, abs = false;
  bool is_float_tok = false, is_hex = false;
  uint64_t imm = 0;  // integer value of the token (sign-extended), or the raw literal
  double fval = 0.0; // value of a float token
  int label = -1;
};

enum Cls : uint8_t { C_VALU, C_SALU, C_BRANCH, C_LDS, C_VMEM, C_SMEM, C_WAIT, C_OTHER, C_N };

struct Inst;
struct Wave;
struct Machine;
typedef void (*ExecFn)(Machine &, Wave &, const Inst &);

struct Inst {
  ExecFn fn = nullptr;
  Cls cls = C_OTHER;
  uint16_t sub = 0;  // sub-operation selector for shared handlers
  Operand o[5];
  uint8_t nops = 0;
  int32_t offset = 0, offset0 = 0, offset1 = 0;
  bool sc0 = false;
  uint32_t bitop3 = 0;
  uint8_t op_sel_hi = 7, op_sel = 0, neg_lo = 0, neg_hi = 0;
  uint8_t sel0 = 6, sel1 = 6, dsel = 6; // SDWA: 0..3 BYTE_n, 4 / 5 WORD_0 / 1, 6 DWORD
  bool dst_preserve = false;
  int line = 0;       // line in the .s file
  uint32_t src_file = 0, src_line = 0; // .loc, when the dump has line tables
  std::string text;
};

struct Kernel {
  std::string name, file;
  std::vector<Inst> code;
  uint32_t lds_static = 0, scratch_bytes = 0, kernarg_size = 0;
  std::vector<std::pair<uint32_t, std::string>> hidden; // (offset in the kernel-argument segment, value kind) of the implicit arguments
  std::vector<uint64_t> hits; // executions per instruction (wave level)
  std::map<uint32_t, std::string> src_files;
};

constexpr uint32_t kSharedHi = 0xFFFF0000u, kPrivateHi = 0xFFFE0000u; // apertures of flat addresses (no host address has these high halves)
constexpr size_t kLdsBytes = 160 * 1024 + 4096;

struct Wave {
  uint32_t s[128];
  uint64_t vcc = 0, exec = 0;
  bool scc = false;
  uint32_t m0 = 0;
  std::vector<uint32_t> v; // [reg][lane]
  std::vector<uint8_t> scratch; // [lane][bytes]
  uint32_t scratch_stride = 0;
  size_t pc = 0;
  bool done = false, at_barrier = false;
  int id = 0;
  inline uint32_t &vr(unsigned r, unsigned lane) { return v[(size_t)r * 64 + lane]; }
};

struct Machine {
  Kernel *k = nullptr;
  std::vector<uint8_t> lds;
  uint32_t block_id = 0;
  uint64_t counts[C_N] = {0};
  bool yield = false;
};

[[noreturn]] static void die(const Inst &in, const char *what) {
  fprintf(stderr, "isa: %s at .s line %d: %s\n", what, in.line, in.text.c_str());
  abort();
}

// ---- value access ----------------------------------------------------------------------------------------------------
static inline uint32_t sreg32(Wave &w, const Operand &o, unsigned idx = 0) {
  switch (o.kind) {
  case K_SGPR: return w.s[o.reg + idx];
  case K_VCC: return idx ? (uint32_t)(w.vcc >> 32) : (uint32_t)w.vcc;
  case K_VCC_LO: return (uint32_t)w.vcc;
  case K_VCC_HI: return (uint32_t)(w.vcc >> 32);
  case K_EXEC: return idx ? (uint32_t)(w.exec >> 32) : (uint32_t)w.exec;
  case K_EXEC_LO: return (uint32_t)w.exec;
  case K_EXEC_HI: return (uint32_t)(w.exec >> 32);
  case K_SCC: return w.scc ? 1u : 0u;
  case K_M0: return w.m0;
  case K_IMM: return idx ? (uint32_t)(o.imm >> 32) : (uint32_t)o.imm;
  case K_SHARED_BASE: return idx ? kSharedHi : 0u;
  case K_PRIVATE_BASE: return idx ? kPrivateHi : 0u;
  case K_NULL: return 0u;
  default: return 0u;
  }
}
static inline uint64_t sreg64(Wave &w, const Operand &o) { return (uint64_t)sreg32(w, o, 0) | ((uint64_t)sreg32(w, o, 1) << 32); }
static inline void swrite32(Wave &w, const Operand &o, uint32_t val, unsigned idx = 0) {
  switch (o.kind) {
  case K_SGPR: w.s[o.reg + idx] = val; break;
  case K_VCC: w.vcc = idx ? ((w.vcc & 0xFFFFFFFFull) | ((uint64_t)val << 32)) : ((w.vcc & ~0xFFFFFFFFull) | val); break;
  case K_VCC_LO: w.vcc = (w.vcc & ~0xFFFFFFFFull) | val; break;
  case K_VCC_HI: w.vcc = (w.vcc & 0xFFFFFFFFull) | ((uint64_t)val << 32); break;
  case K_EXEC: w.exec = idx ? ((w.exec & 0xFFFFFFFFull) | ((uint64_t)val << 32)) : ((w.exec & ~0xFFFFFFFFull) | val); break;
  case K_EXEC_LO: w.exec = (w.exec & ~0xFFFFFFFFull) | val; break;
  case K_EXEC_HI: w.exec = (w.exec & 0xFFFFFFFFull) | ((uint64_t)val << 32); break;
  case K_M0: w.m0 = val; break;
  case K_NULL: break;
  default: fprintf(stderr, "isa: scalar write to operand kind %d\n", (int)o.kind); abort();
  }
}
static inline void swrite64(Wave &w, const Operand &o, uint64_t val) {
  if (o.kind == K_VCC) w.vcc = val;
  else if (o.kind == K_EXEC) w.exec = val;
  else {
    swrite32(w, o, (uint32_t)val, 0);
    swrite32(w, o, (uint32_t)(val >> 32), 1);
  }
}
// a 32-bit source of a vector instruction for `lane` (word `idx` of a multi-register operand)
static inline uint32_t src32(Wave &w, const Operand &o, unsigned lane, unsigned idx = 0) {
  if (o.kind == K_VGPR) return w.vr(o.reg + idx, lane);
  return sreg32(w, o, idx);
}
static inline uint64_t src64(Wave &w, const Operand &o, unsigned lane) {
  if (o.kind == K_VGPR) return (uint64_t)w.vr(o.reg, lane) | ((uint64_t)w.vr(o.reg + 1, lane) << 32);
  return sreg64(w, o);
}
static inline double as_f64(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static inline uint64_t f64_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }
static inline float as_f32(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t f32_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
static inline double srcd(Wave &w, const Operand &o, unsigned lane) {
  uint64_t b = src64(w, o, lane);
  if (o.abs) b &= ~(1ull << 63);
  if (o.neg) b ^= (1ull << 63);
  return as_f64(b);
}
static inline float srcf(Wave &w, const Operand &o, unsigned lane, unsigned idx = 0) {
  uint32_t b = src32(w, o, lane, idx);
  if (o.abs) b &= 0x7FFFFFFFu;
  if (o.neg) b ^= 0x80000000u;
  return as_f32(b);
}
static inline void dst64(Wave &w, const Operand &o, unsigned lane, uint64_t v) {
  w.vr(o.reg, lane) = (uint32_t)v;
  w.vr(o.reg + 1, lane) = (uint32_t)(v >> 32);
}
static inline void dstd(Wave &w, const Operand &o, unsigned lane, double d) { dst64(w, o, lane, f64_bits(d)); }
#define FOR_LANES(w) for (uint64_t m_ = (w).exec; m_; m_ &= m_ - 1) for (unsigned lane = (unsigned)__builtin_ctzll(m_), once_ = 1; once_; once_ = 0)

// ---- memory ------------------------------------------------------------------------------------------------------------
static inline uint8_t *flat_ptr(Machine &M, Wave &w, unsigned lane, uint64_t addr, size_t bytes, const Inst &in) {
  const uint32_t hi = (uint32_t)(addr >> 32);
  if (hi == kSharedHi) {
    const uint32_t a = (uint32_t)addr;
    if ((size_t)a + bytes > M.lds.size()) die(in, "flat access beyond LDS");
    return M.lds.data() + a;
  }
  if (hi == kPrivateHi) {
    const uint32_t a = (uint32_t)addr;
    if ((size_t)a + bytes > w.scratch_stride) die(in, "flat access beyond scratch");
    return w.scratch.data() + (size_t)lane * w.scratch_stride + a;
  }
  if (addr < 4096) die(in, "flat / global access to a null page address");
  return reinterpret_cast<uint8_t *>(addr);
}
static inline uint8_t *lds_ptr(Machine &M, uint32_t a, size_t bytes, const Inst &in) {
  if ((size_t)a + bytes > M.lds.size()) die(in, "LDS access out of range");
  return M.lds.data() + a;
}

// ---------------------------------------------------------------------------------------------------------------- handlers
// sub codes for typed families
enum { T_F64, T_F32, T_I32, T_U32, T_U64, T_I64, T_U16, T_I16 };
enum { P_F, P_LT, P_EQ, P_LE, P_GT, P_LG, P_GE, P_O, P_U, P_NGE, P_NLG, P_NGT, P_NLE, P_NEQ, P_NLT, P_T, P_NE /*int*/ };

template <typename T> static inline bool cmp_pred(int p, T a, T b) {
  switch (p) {
  case P_LT: return a < b;
  case P_EQ: return a == b;
  case P_LE: return a <= b;
  case P_GT: return a > b;
  case P_LG: return a < b || a > b;
  case P_GE: return a >= b;
  case P_O: return a == a && b == b;
  case P_U: return !(a == a && b == b);
  case P_NGE: return !(a >= b);
  case P_NLG: return !(a < b || a > b);
  case P_NGT: return !(a > b);
  case P_NLE: return !(a <= b);
  case P_NEQ: return !(a == b);
  case P_NLT: return !(a < b);
  case P_NE: return a != b;
  case P_T: return true;
  default: return false;
  }
}
// v_cmp_<pred>_<type>: o[0] = sdst (vcc or an SGPR pair), o[1], o[2]; sub = type << 8 | pred
static void x_vcmp(Machine &, Wave &w, const Inst &in) {
  const int ty = in.sub >> 8, p = in.sub & 0xFF;
  uint64_t res = 0;
  FOR_LANES(w) {
    bool r = false;
    switch (ty) {
    case T_F64: r = cmp_pred<double>(p, srcd(w, in.o[1], lane), srcd(w, in.o[2], lane)); break;
    case T_F32: r = cmp_pred<float>(p, srcf(w, in.o[1], lane), srcf(w, in.o[2], lane)); break;
    case T_I32: r = cmp_pred<int32_t>(p, (int32_t)src32(w, in.o[1], lane), (int32_t)src32(w, in.o[2], lane)); break;
    case T_U32: r = cmp_pred<uint32_t>(p, src32(w, in.o[1], lane), src32(w, in.o[2], lane)); break;
    case T_U16: r = cmp_pred<uint32_t>(p, src32(w, in.o[1], lane) & 0xFFFFu, src32(w, in.o[2], lane) & 0xFFFFu); break;
    case T_I16: r = cmp_pred<int32_t>(p, (int32_t)(int16_t)src32(w, in.o[1], lane), (int32_t)(int16_t)src32(w, in.o[2], lane)); break;
    case T_U64: r = cmp_pred<uint64_t>(p, src64(w, in.o[1], lane), src64(w, in.o[2], lane)); break;
    case T_I64: r = cmp_pred<int64_t>(p, (int64_t)src64(w, in.o[1], lane), (int64_t)src64(w, in.o[2], lane)); break;
    }
    if (r) res |= 1ull << lane;
  }
  swrite64(w, in.o[0], res); // inactive lanes read 0
}
static uint32_t fclass64(double d) {
  const uint64_t b = f64_bits(d);
  const bool neg = b >> 63;
  const uint64_t e = (b >> 52) & 0x7FF, m = b & 0xFFFFFFFFFFFFFull;
  if (e == 0x7FF && m) return (m >> 51) ? 2u : 1u; // qNaN : sNaN
  if (e == 0x7FF) return neg ? 4u : 512u;
  if (e == 0 && m == 0) return neg ? 32u : 64u;
  if (e == 0) return neg ? 16u : 128u;
  return neg ? 8u : 256u;
}
static uint32_t fclass32(float d) {
  const uint32_t b = f32_bits(d);
  const bool neg = b >> 31;
  const uint32_t e = (b >> 23) & 0xFF, m = b & 0x7FFFFFu;
  if (e == 0xFF && m) return (m >> 22) ? 2u : 1u;
  if (e == 0xFF) return neg ? 4u : 512u;
  if (e == 0 && m == 0) return neg ? 32u : 64u;
  if (e == 0) return neg ? 16u : 128u;
  return neg ? 8u : 256u;
}
static void x_vcmp_class_f32(Machine &, Wave &w, const Inst &in) {
  uint64_t res = 0;
  FOR_LANES(w) {
    if (fclass32(srcf(w, in.o[1], lane)) & src32(w, in.o[2], lane)) res |= 1ull << lane;
  }
  swrite64(w, in.o[0], res);
}
static void x_vcmp_class_f64(Machine &, Wave &w, const Inst &in) {
  uint64_t res = 0;
  FOR_LANES(w) {
    if (fclass64(srcd(w, in.o[1], lane)) & src32(w, in.o[2], lane)) res |= 1ull << lane;
  }
  swrite64(w, in.o[0], res);
}

static inline double min_f64(double a, double b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == 0.0 && b == 0.0) return std::signbit(a) ? a : b;
  return a < b ? a : b;
}
static inline double max_f64(double a, double b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == 0.0 && b == 0.0) return std::signbit(a) ? b : a;
  return a > b ? a : b;
}
static inline float min_f32(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == 0.0f && b == 0.0f) return std::signbit(a) ? a : b;
  return a < b ? a : b;
}
static inline float max_f32(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == 0.0f && b == 0.0f) return std::signbit(a) ? b : a;
  return a > b ? a : b;
}
static inline double quiet(double d) { return d != d ? as_f64(f64_bits(d) | (1ull << 51)) : d; }
static inline float quietf(float d) { return d != d ? as_f32(f32_bits(d) | (1u << 22)) : d; }

enum { D_ADD, D_MUL, D_FMA, D_FMAC, D_MIN, D_MAX, D_LDEXP, D_RCP, D_RSQ, D_FRACT, D_RNDNE, D_MOV64, D_TRUNC, D_FLOOR, D_FREXP_MANT, D_CEIL, D_SQRT };
static void x_f64(Machine &, Wave &w, const Inst &in) {
  FOR_LANES(w) {
    double r = 0.0;
    switch (in.sub) {
    case D_ADD: r = srcd(w, in.o[1], lane) + srcd(w, in.o[2], lane); break;
    case D_MUL: r = srcd(w, in.o[1], lane) * srcd(w, in.o[2], lane); break;
    case D_FMA: r = std::fma(srcd(w, in.o[1], lane), srcd(w, in.o[2], lane), srcd(w, in.o[3], lane)); break;
    case D_FMAC: r = std::fma(srcd(w, in.o[1], lane), srcd(w, in.o[2], lane), as_f64(src64(w, in.o[0], lane))); break;
    case D_MIN: r = quiet(min_f64(srcd(w, in.o[1], lane), srcd(w, in.o[2], lane))); break;
    case D_MAX: r = quiet(max_f64(srcd(w, in.o[1], lane), srcd(w, in.o[2], lane))); break;
    case D_LDEXP: r = std::ldexp(srcd(w, in.o[1], lane), (int)(int32_t)src32(w, in.o[2], lane)); break;
    case D_RCP: r = 1.0 / srcd(w, in.o[1], lane); break;
    case D_RSQ: {
      const double x = srcd(w, in.o[1], lane);
      r = (double)(1.0L / sqrtl((long double)x));
      if (x == 0.0) r = std::signbit(x) ? -INFINITY : INFINITY;
      break;
    }
    case D_FRACT: {
      const double x = srcd(w, in.o[1], lane);
      r = x - std::floor(x);
      if (!(r < 1.0)) r = as_f64(0x3FEFFFFFFFFFFFFFull);
      if (x != x) r = x;
      if (std::isinf(x)) r = NAN;
      break;
    }
    case D_RNDNE: r = std::nearbyint(srcd(w, in.o[1], lane)); break;
    case D_TRUNC: r = std::trunc(srcd(w, in.o[1], lane)); break;
    case D_CEIL: r = std::ceil(srcd(w, in.o[1], lane)); break;
    case D_SQRT: r = std::sqrt(srcd(w, in.o[1], lane)); break;
    case D_FREXP_MANT: { const double x = srcd(w, in.o[1], lane); int e; r = (std::isinf(x) || x != x) ? x : std::frexp(x, &e); break; }
    case D_FLOOR: r = std::floor(srcd(w, in.o[1], lane)); break;
    }
    dstd(w, in.o[0], lane, r);
  }
}
enum { F_ADD, F_SUB, F_MUL, F_FMA, F_FMAC, F_MIN, F_MAX, F_MIN3, F_MAX3, F_RCP, F_SQRT, F_TRUNC, F_FLOOR, F_RSQ };
static void x_f32(Machine &, Wave &w, const Inst &in) {
  FOR_LANES(w) {
    float r = 0.0f;
    switch (in.sub) {
    case F_ADD: r = srcf(w, in.o[1], lane) + srcf(w, in.o[2], lane); break;
    case F_SUB: r = srcf(w, in.o[1], lane) - srcf(w, in.o[2], lane); break;
    case F_MUL: r = srcf(w, in.o[1], lane) * srcf(w, in.o[2], lane); break;
    case F_FMA: r = std::fmaf(srcf(w, in.o[1], lane), srcf(w, in.o[2], lane), srcf(w, in.o[3], lane)); break;
    case F_FMAC: r = std::fmaf(srcf(w, in.o[1], lane), srcf(w, in.o[2], lane), as_f32(w.vr(in.o[0].reg, lane))); break;
    case F_MIN: r = quietf(min_f32(srcf(w, in.o[1], lane), srcf(w, in.o[2], lane))); break;
    case F_MAX: r = quietf(max_f32(srcf(w, in.o[1], lane), srcf(w, in.o[2], lane))); break;
    case F_MIN3: r = quietf(min_f32(min_f32(srcf(w, in.o[1], lane), srcf(w, in.o[2], lane)), srcf(w, in.o[3], lane))); break;
    case F_MAX3: r = quietf(max_f32(max_f32(srcf(w, in.o[1], lane), srcf(w, in.o[2], lane)), srcf(w, in.o[3], lane))); break;
    case F_RCP: r = 1.0f / srcf(w, in.o[1], lane); break;
    case F_SQRT: r = std::sqrt(srcf(w, in.o[1], lane)); break;
    case F_TRUNC: r = std::trunc(srcf(w, in.o[1], lane)); break;
    case F_RSQ: r = (float)(1.0 / std::sqrt((double)srcf(w, in.o[1], lane))); break;
    case F_FLOOR: r = std::floor(srcf(w, in.o[1], lane)); break;
    }
    w.vr(in.o[0].reg, lane) = f32_bits(r);
  }
}
// v_pk_{mul,fma}_f32: two floats per operand; op_sel / op_sel_hi pick the word of each SOURCE used for the low / high result
static void x_pk_f32(Machine &, Wave &w, const Inst &in) {
  const int nsrc = in.sub == F_FMA ? 3 : 2;
  FOR_LANES(w) {
    float r[2];
    for (int half = 0; half < 2; ++half) {
      float s[3] = {0, 0, 0};
      for (int k = 0; k < nsrc; ++k) {
        const unsigned sel = half ? ((in.op_sel_hi >> k) & 1u) : ((in.op_sel >> k) & 1u);
        const Operand &o = in.o[1 + k];
        s[k] = srcf(w, o, lane, (o.kind == K_VGPR || o.kind == K_SGPR) ? sel : 0u);
        if (((half ? in.neg_hi : in.neg_lo) >> k) & 1u) s[k] = -s[k];
      }
      r[half] = in.sub == F_FMA ? std::fmaf(s[0], s[1], s[2]) : (in.sub == F_ADD ? s[0] + s[1] : s[0] * s[1]);
    }
    w.vr(in.o[0].reg, lane) = f32_bits(r[0]);
    w.vr(in.o[0].reg + 1, lane) = f32_bits(r[1]);
  }
}

// ---- division helpers (ISA pseudo-code) ---------------------------------------------------------------------------------
static inline int exp64(double d) { return (int)((f64_bits(d) >> 52) & 0x7FF); }
static inline bool denorm64(double d) { return exp64(d) == 0 && (f64_bits(d) & 0xFFFFFFFFFFFFFull); }
static inline int exp32(float d) { return (int)((f32_bits(d) >> 23) & 0xFF); }
static inline bool denorm32(float d) { return exp32(d) == 0 && (f32_bits(d) & 0x7FFFFFu); }
// v_div_scale_f64 vdst, sdst(vcc), s0, s1 (denominator), s2 (numerator)
static void x_div_scale_f64(Machine &, Wave &w, const Inst &in) {
  uint64_t vcc = 0;
  FOR_LANES(w) {
    const double s0 = srcd(w, in.o[2], lane), s1 = srcd(w, in.o[3], lane), s2 = srcd(w, in.o[4], lane);
    double d = s0;
    bool flag = false;
    if (s2 == 0.0 || s1 == 0.0) d = NAN;
    else if (exp64(s2) - exp64(s1) >= 768) {
      flag = true;
      if (f64_bits(s0) == f64_bits(s1)) d = std::ldexp(s0, 128);
    } else if (denorm64(s1)) d = std::ldexp(s0, 128);
    else if (denorm64(1.0 / s1) && denorm64(s2 / s1)) {
      flag = true;
      if (f64_bits(s0) == f64_bits(s1)) d = std::ldexp(s0, 128);
    } else if (denorm64(1.0 / s1)) d = std::ldexp(s0, -128);
    else if (denorm64(s2 / s1)) {
      flag = true;
      if (f64_bits(s0) == f64_bits(s2)) d = std::ldexp(s0, 128);
    } else if (exp64(s2) <= 53) d = std::ldexp(s0, 128);
    if (flag) vcc |= 1ull << lane;
    dstd(w, in.o[0], lane, d);
  }
  swrite64(w, in.o[1], vcc);
}
static void x_div_scale_f32(Machine &, Wave &w, const Inst &in) {
  uint64_t vcc = 0;
  FOR_LANES(w) {
    const float s0 = srcf(w, in.o[2], lane), s1 = srcf(w, in.o[3], lane), s2 = srcf(w, in.o[4], lane);
    float d = s0;
    bool flag = false;
    if (s2 == 0.0f || s1 == 0.0f) d = NAN;
    else if (exp32(s2) - exp32(s1) >= 96) {
      flag = true;
      if (f32_bits(s0) == f32_bits(s1)) d = std::ldexp(s0, 64);
    } else if (denorm32(s1)) d = std::ldexp(s0, 64);
    else if (denorm32(1.0f / s1) && denorm32(s2 / s1)) {
      flag = true;
      if (f32_bits(s0) == f32_bits(s1)) d = std::ldexp(s0, 64);
    } else if (denorm32(1.0f / s1)) d = std::ldexp(s0, -64);
    else if (denorm32(s2 / s1)) {
      flag = true;
      if (f32_bits(s0) == f32_bits(s2)) d = std::ldexp(s0, 64);
    } else if (exp32(s2) <= 23) d = std::ldexp(s0, 64);
    if (flag) vcc |= 1ull << lane;
    w.vr(in.o[0].reg, lane) = f32_bits(d);
  }
  swrite64(w, in.o[1], vcc);
}
static void x_div_fmas_f64(Machine &, Wave &w, const Inst &in) {
  FOR_LANES(w) {
    double r = std::fma(srcd(w, in.o[1], lane), srcd(w, in.o[2], lane), srcd(w, in.o[3], lane));
    if ((w.vcc >> lane) & 1) r = std::ldexp(r, 64);
    dstd(w, in.o[0], lane, r);
  }
}
static void x_div_fmas_f32(Machine &, Wave &w, const Inst &in) {
  FOR_LANES(w) {
    float r = std::fmaf(srcf(w, in.o[1], lane), srcf(w, in.o[2], lane), srcf(w, in.o[3], lane));
    if ((w.vcc >> lane) & 1) r = std::ldexp(r, 32);
    w.vr(in.o[0].reg, lane) = f32_bits(r);
  }
}
// v_div_fixup vdst, s0 (quotient), s1 (denominator), s2 (numerator)
static void x_div_fixup_f64(Machine &, Wave &w, const Inst &in) {
  FOR_LANES(w) {
    const double s0 = srcd(w, in.o[1], lane), s1 = srcd(w, in.o[2], lane), s2 = srcd(w, in.o[3], lane);
    const bool sign = std::signbit(s1) != std::signbit(s2);
    double d;
    if (s2 != s2) d = quiet(s2);
    else if (s1 != s1) d = quiet(s1);
    else if (s1 == 0.0 && s2 == 0.0) d = as_f64(0xFFF8000000000000ull);
    else if (std::isinf(s1) && std::isinf(s2)) d = as_f64(0xFFF8000000000000ull);
    else if (s1 == 0.0 || std::isinf(s2)) d = sign ? -INFINITY : INFINITY;
    else if (std::isinf(s1) || s2 == 0.0) d = sign ? -0.0 : 0.0;
    else if (exp64(s2) - exp64(s1) < -1075) d = sign ? -0.0 : 0.0;
    else if (exp64(s1) == 2047) d = sign ? -INFINITY : INFINITY;
    else d = sign ? -std::fabs(s0) : std::fabs(s0);
    dstd(w, in.o[0], lane, d);
  }
}
static void x_div_fixup_f32(Machine &, Wave &w, const Inst &in) {
  FOR_LANES(w) {
    const float s0 = srcf(w, in.o[1], lane), s1 = srcf(w, in.o[2], lane), s2 = srcf(w, in.o[3], lane);
    const bool sign = std::signbit(s1) != std::signbit(s2);
    float d;
    if (s2 != s2) d = quietf(s2);
    else if (s1 != s1) d = quietf(s1);
    else if (s1 == 0.0f && s2 == 0.0f) d = as_f32(0xFFC00000u);
    else if (std::isinf(s1) && std::isinf(s2)) d = as_f32(0xFFC00000u);
    else if (s1 == 0.0f || std::isinf(s2)) d = sign ? -INFINITY : INFINITY;
    else if (std::isinf(s1) || s2 == 0.0f) d = sign ? -0.0f : 0.0f;
    else if (exp32(s2) - exp32(s1) < -150) d = sign ? -0.0f : 0.0f;
    else if (exp32(s1) == 255) d = sign ? -INFINITY : INFINITY;
    else d = sign ? -std::fabs(s0) : std::fabs(s0);
    w.vr(in.o[0].reg, lane) = f32_bits(d);
  }
}

// ---- conversions -----------------------------------------------------------------------------------------------------
enum { CV_F32_F64, CV_F64_F32, CV_F64_I32, CV_F64_U32, CV_I32_F64, CV_U32_F64, CV_F32_I32, CV_F32_U32, CV_U32_F32, CV_I32_F32, CV_FREXP_EXP_F64 };
static void x_cvt(Machine &, Wave &w, const Inst &in) {
  FOR_LANES(w) {
    switch (in.sub) {
    case CV_F32_F64: w.vr(in.o[0].reg, lane) = f32_bits((float)srcd(w, in.o[1], lane)); break;
    case CV_F64_F32: dstd(w, in.o[0], lane, (double)srcf(w, in.o[1], lane)); break;
    case CV_F64_I32: dstd(w, in.o[0], lane, (double)(int32_t)src32(w, in.o[1], lane)); break;
    case CV_F64_U32: dstd(w, in.o[0], lane, (double)src32(w, in.o[1], lane)); break;
    case CV_I32_F64: {
      const double x = srcd(w, in.o[1], lane);
      int32_t r = x != x ? 0 : (x >= 2147483647.0 ? INT32_MAX : (x <= -2147483648.0 ? INT32_MIN : (int32_t)x));
      w.vr(in.o[0].reg, lane) = (uint32_t)r;
      break;
    }
    case CV_U32_F64: {
      const double x = srcd(w, in.o[1], lane);
      w.vr(in.o[0].reg, lane) = x != x || x <= 0.0 ? 0u : (x >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)x);
      break;
    }
    case CV_FREXP_EXP_F64: {
      const double x = srcd(w, in.o[1], lane);
      int e = 0;
      if (!(std::isinf(x) || x != x || x == 0.0)) (void)std::frexp(x, &e);
      w.vr(in.o[0].reg, lane) = (uint32_t)e;
      break;
    }
    case CV_F32_I32: w.vr(in.o[0].reg, lane) = f32_bits((float)(int32_t)src32(w, in.o[1], lane)); break;
    case CV_F32_U32: w.vr(in.o[0].reg, lane) = f32_bits((float)src32(w, in.o[1], lane)); break;
    case CV_U32_F32: {
      const float x = srcf(w, in.o[1], lane);
      w.vr(in.o[0].reg, lane) = x != x || x <= 0.0f ? 0u : (x >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)x);
      break;
    }
    case CV_I32_F32: {
      const float x = srcf(w, in.o[1], lane);
      int32_t r = x != x ? 0 : (x >= 2147483648.0f ? INT32_MAX : (x <= -2147483648.0f ? INT32_MIN : (int32_t)x));
      w.vr(in.o[0].reg, lane) = (uint32_t)r;
      break;
    }
    }
  }
}

// ---- 32 / 64-bit integer vector ops -----------------------------------------------------------------------------------
enum {
  I_MOV, I_MOV64, I_ADD, I_SUB, I_SUBREV, I_AND, I_OR, I_XOR, I_NOT, I_LSHLREV, I_LSHRREV, I_ASHRREV, I_LSHLREV64, I_LSHRREV64, I_ADD3, I_OR3, I_AND_OR, I_LSHL_ADD,
  I_ADD_LSHL, I_LSHL_OR, I_LSHL_ADD64, I_BFE_U32, I_BFREV, I_BCNT, I_MUL_LO, I_MUL_HI, I_MUL_U24, I_MAD_U24, I_MIN_I32, I_MIN_U32, I_MAX_I32, I_MAX_U32, I_MBCNT_LO, I_MBCNT_HI,
  I_BITOP3, I_CNDMASK, I_ALIGNBIT, I_MUL_HI_U24, I_FFBL, I_FFBH, I_MAD_U16, I_MUL_I24, I_MAD_I24, I_XAD, I_ADD_U16
};
static void x_int(Machine &, Wave &w, const Inst &in) {
  FOR_LANES(w) {
    const Operand *o = in.o;
    uint32_t r = 0;
    switch (in.sub) {
    case I_MOV: r = src32(w, o[1], lane); break;
    case I_MOV64: dst64(w, o[0], lane, src64(w, o[1], lane)); continue;
    case I_ADD: r = src32(w, o[1], lane) + src32(w, o[2], lane); break;
    case I_SUB: r = src32(w, o[1], lane) - src32(w, o[2], lane); break;
    case I_SUBREV: r = src32(w, o[2], lane) - src32(w, o[1], lane); break;
    case I_AND: r = src32(w, o[1], lane) & src32(w, o[2], lane); break;
    case I_OR: r = src32(w, o[1], lane) | src32(w, o[2], lane); break;
    case I_XOR: r = src32(w, o[1], lane) ^ src32(w, o[2], lane); break;
    case I_NOT: r = ~src32(w, o[1], lane); break;
    case I_LSHLREV: r = src32(w, o[2], lane) << (src32(w, o[1], lane) & 31u); break;
    case I_LSHRREV: r = src32(w, o[2], lane) >> (src32(w, o[1], lane) & 31u); break;
    case I_ASHRREV: r = (uint32_t)((int32_t)src32(w, o[2], lane) >> (src32(w, o[1], lane) & 31u)); break;
    case I_LSHLREV64: dst64(w, o[0], lane, src64(w, o[2], lane) << (src32(w, o[1], lane) & 63u)); continue;
    case I_LSHRREV64: dst64(w, o[0], lane, src64(w, o[2], lane) >> (src32(w, o[1], lane) & 63u)); continue;
    case I_ADD3: r = src32(w, o[1], lane) + src32(w, o[2], lane) + src32(w, o[3], lane); break;
    case I_OR3: r = src32(w, o[1], lane) | src32(w, o[2], lane) | src32(w, o[3], lane); break;
    case I_AND_OR: r = (src32(w, o[1], lane) & src32(w, o[2], lane)) | src32(w, o[3], lane); break;
    case I_LSHL_ADD: r = (src32(w, o[1], lane) << (src32(w, o[2], lane) & 31u)) + src32(w, o[3], lane); break;
    case I_ADD_LSHL: r = (src32(w, o[1], lane) + src32(w, o[2], lane)) << (src32(w, o[3], lane) & 31u); break;
    case I_LSHL_OR: r = (src32(w, o[1], lane) << (src32(w, o[2], lane) & 31u)) | src32(w, o[3], lane); break;
    case I_LSHL_ADD64: dst64(w, o[0], lane, (src64(w, o[1], lane) << (src32(w, o[2], lane) & 7u)) + src64(w, o[3], lane)); continue;
    case I_BFE_U32: {
      const uint32_t off = src32(w, o[2], lane) & 31u, wd = src32(w, o[3], lane) & 31u;
      r = wd == 0 ? 0u : ((src32(w, o[1], lane) >> off) & ((1u << wd) - 1u));
      break;
    }
    case I_BFREV: {
      uint32_t x = src32(w, o[1], lane);
      for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i);
      break;
    }
    case I_BCNT: r = (uint32_t)__builtin_popcount(src32(w, o[1], lane)) + src32(w, o[2], lane); break;
    case I_MUL_LO: r = src32(w, o[1], lane) * src32(w, o[2], lane); break;
    case I_MUL_HI: r = (uint32_t)(((uint64_t)src32(w, o[1], lane) * (uint64_t)src32(w, o[2], lane)) >> 32); break;
    case I_MUL_U24: r = (src32(w, o[1], lane) & 0xFFFFFFu) * (src32(w, o[2], lane) & 0xFFFFFFu); break;
    case I_MAD_U24: r = (src32(w, o[1], lane) & 0xFFFFFFu) * (src32(w, o[2], lane) & 0xFFFFFFu) + src32(w, o[3], lane); break;
    case I_MIN_I32: r = (uint32_t)std::min((int32_t)src32(w, o[1], lane), (int32_t)src32(w, o[2], lane)); break;
    case I_MAX_I32: r = (uint32_t)std::max((int32_t)src32(w, o[1], lane), (int32_t)src32(w, o[2], lane)); break;
    case I_MIN_U32: r = std::min(src32(w, o[1], lane), src32(w, o[2], lane)); break;
    case I_MAX_U32: r = std::max(src32(w, o[1], lane), src32(w, o[2], lane)); break;
    case I_MBCNT_LO: {
      const uint32_t mask = lane >= 32 ? 0xFFFFFFFFu : ((1u << lane) - 1u);
      r = (uint32_t)__builtin_popcount(src32(w, o[1], lane) & mask) + src32(w, o[2], lane);
      break;
    }
    case I_MBCNT_HI: {
      const uint32_t mask = lane < 32 ? 0u : (lane == 32 ? 0u : ((1u << (lane - 32)) - 1u));
      r = (uint32_t)__builtin_popcount(src32(w, o[1], lane) & mask) + src32(w, o[2], lane);
      break;
    }
    case I_BITOP3: {
      const uint32_t a = src32(w, o[1], lane), b = src32(w, o[2], lane), c = src32(w, o[3], lane);
      for (int i = 0; i < 32; ++i) {
        const unsigned idx = (((a >> i) & 1u) << 2) | (((b >> i) & 1u) << 1) | ((c >> i) & 1u);
        r |= ((in.bitop3 >> idx) & 1u) << i;
      }
      break;
    }
    case I_ALIGNBIT: r = (uint32_t)(((((uint64_t)src32(w, o[1], lane)) << 32) | src32(w, o[2], lane)) >> (src32(w, o[3], lane) & 31u)); break;
    case I_MUL_HI_U24: r = (uint32_t)(((uint64_t)(src32(w, o[1], lane) & 0xFFFFFFu) * (uint64_t)(src32(w, o[2], lane) & 0xFFFFFFu)) >> 32); break;
    case I_FFBL: { const uint32_t x = src32(w, o[1], lane); r = x ? (uint32_t)__builtin_ctz(x) : 0xFFFFFFFFu; break; }
    case I_FFBH: { const uint32_t x = src32(w, o[1], lane); r = x ? (uint32_t)__builtin_clz(x) : 0xFFFFFFFFu; break; }
    case I_MAD_U16: r = (src32(w, o[1], lane) & 0xFFFFu) * (src32(w, o[2], lane) & 0xFFFFu) + src32(w, o[3], lane); break;
    case I_MUL_I24: r = (uint32_t)(((int32_t)(src32(w, o[1], lane) << 8) >> 8) * ((int32_t)(src32(w, o[2], lane) << 8) >> 8)); break;
    case I_MAD_I24: r = (uint32_t)(((int32_t)(src32(w, o[1], lane) << 8) >> 8) * ((int32_t)(src32(w, o[2], lane) << 8) >> 8)) + src32(w, o[3], lane); break;
    case I_ADD_U16: r = (src32(w, o[1], lane) + src32(w, o[2], lane)) & 0xFFFFu; break;
    case I_XAD: r = (src32(w, o[1], lane) ^ src32(w, o[2], lane)) + src32(w, o[3], lane); break;
    case I_CNDMASK: { // o[1] when the lane's mask bit is 0, o[2] when 1; float modifiers apply
      const uint64_t mask = in.nops > 3 ? sreg64(w, o[3]) : w.vcc;
      const Operand &p = ((mask >> lane) & 1) ? o[2] : o[1];
      r = src32(w, p, lane);
      if (p.abs) r &= 0x7FFFFFFFu;
      if (p.neg) r ^= 0x80000000u;
      break;
    }
    }
    w.vr(o[0].reg, lane) = r;
  }
}
// carry ops: v_add_co_u32 vdst, sdst, a, b ; v_addc_co_u32 vdst, sdst, a, b, ssrc ; v_subbrev_co_u32 vdst, sdst, a, b, ssrc  (b - a - borrow)
enum { CO_ADD, CO_ADDC, CO_SUBBREV, CO_SUB, CO_SUBB, CO_SUBREV };
static void x_carry(Machine &, Wave &w, const Inst &in) {
  const uint64_t cin = in.nops > 4 ? sreg64(w, in.o[4]) : 0;
  uint64_t cout = 0;
  FOR_LANES(w) {
    const uint64_t a = src32(w, in.o[2], lane), b = src32(w, in.o[3], lane), c = (cin >> lane) & 1;
    uint64_t r = 0;
    bool carry = false;
    switch (in.sub) {
    case CO_ADD: r = a + b; carry = r >> 32; break;
    case CO_ADDC: r = a + b + c; carry = r >> 32; break;
    case CO_SUB: r = a - b; carry = b > a; break;
    case CO_SUBB: r = a - b - c; carry = b + c > a; break;
    case CO_SUBBREV: r = b - a - c; carry = a + c > b; break;
    case CO_SUBREV: r = b - a; carry = a > b; break;
    }
    if (carry) cout |= 1ull << lane;
    w.vr(in.o[0].reg, lane) = (uint32_t)r;
  }
  // sdst gets the carries of the active lanes, the other bits keep ... the ISA writes 0 for inactive lanes of a VOP3 sdst
  swrite64(w, in.o[1], cout);
}
// v_mad_u64_u32 vdst(64), sdst, a, b, c(64) ; v_mad_i64_i32
static void x_mad64(Machine &, Wave &w, const Inst &in) {
  uint64_t cout = 0;
  FOR_LANES(w) {
    const uint64_t c = src64(w, in.o[4], lane);
    uint64_t r;
    if (in.sub == 0) {
      const unsigned __int128 t = (unsigned __int128)src32(w, in.o[2], lane) * src32(w, in.o[3], lane) + c;
      r = (uint64_t)t;
      if (t >> 64) cout |= 1ull << lane;
    } else {
      r = (uint64_t)((int64_t)(int32_t)src32(w, in.o[2], lane) * (int64_t)(int32_t)src32(w, in.o[3], lane) + (int64_t)c);
    }
    dst64(w, in.o[0], lane, r);
  }
  swrite64(w, in.o[1], cout);
}
// v_pk_mov_b32 vdst[2], src0[2], src1[2] op_sel:[a,b]: D.lo = src0[a], D.hi = src1[b]
static void x_pk_mov(Machine &, Wave &w, const Inst &in) {
  FOR_LANES(w) {
    const uint32_t lo = src32(w, in.o[1], lane, in.op_sel & 1u), hi = src32(w, in.o[2], lane, (in.op_sel >> 1) & 1u);
    w.vr(in.o[0].reg, lane) = lo;
    w.vr(in.o[0].reg + 1, lane) = hi;
  }
}
// SDWA forms (sub-dword source / destination selection) of the few VOP2 / VOPC operations the kernels use
static inline uint32_t sdwa_sel(uint32_t v, unsigned sel) { return sel < 4 ? ((v >> (8 * sel)) & 0xFFu) : (sel < 6 ? ((v >> (16 * (sel - 4))) & 0xFFFFu) : v); }
enum { SD_ADD_U32, SD_MIN_U16, SD_OR_B32, SD_CMP_GT_U16, SD_CMP_NE_U16, SD_CMP_EQ_U16, SD_CMP_LT_U16, SD_AND_B32, SD_MOV_B32, SD_LSHLREV_B32 };
static void x_sdwa(Machine &, Wave &w, const Inst &in) {
  const bool is_cmp = in.sub >= SD_CMP_GT_U16 && in.sub <= SD_CMP_LT_U16;
  uint64_t res = 0;
  FOR_LANES(w) {
    const uint32_t a = sdwa_sel(src32(w, in.o[1], lane), in.sel0), b = in.nops > 2 ? sdwa_sel(src32(w, in.o[2], lane), in.sel1) : 0u;
    uint32_t r = 0;
    switch (in.sub) {
    case SD_ADD_U32: r = a + b; break;
    case SD_MIN_U16: r = std::min(a & 0xFFFFu, b & 0xFFFFu); break;
    case SD_OR_B32: r = a | b; break;
    case SD_AND_B32: r = a & b; break;
    case SD_MOV_B32: r = a; break;
    case SD_LSHLREV_B32: r = b << (a & 31u); break;
    case SD_CMP_GT_U16: r = (a & 0xFFFFu) > (b & 0xFFFFu); break;
    case SD_CMP_LT_U16: r = (a & 0xFFFFu) < (b & 0xFFFFu); break;
    case SD_CMP_NE_U16: r = (a & 0xFFFFu) != (b & 0xFFFFu); break;
    case SD_CMP_EQ_U16: r = (a & 0xFFFFu) == (b & 0xFFFFu); break;
    }
    if (is_cmp) {
      if (r) res |= 1ull << lane;
      continue;
    }
    uint32_t &d = w.vr(in.o[0].reg, lane);
    if (in.dsel == 6) d = r;
    else {
      const unsigned sh = in.dsel < 4 ? 8 * in.dsel : 16 * (in.dsel - 4);
      const uint32_t mask = (in.dsel < 4 ? 0xFFu : 0xFFFFu) << sh;
      d = ((in.dst_preserve ? d : 0u) & ~mask) | ((r << sh) & mask);
    }
  }
  if (is_cmp) swrite64(w, in.o[0], res);
}
static void x_readlane(Machine &, Wave &w, const Inst &in) { swrite32(w, in.o[0], w.vr(in.o[1].reg, sreg32(w, in.o[2]) & 63u)); }
static void x_writelane(Machine &, Wave &w, const Inst &in) { w.vr(in.o[0].reg, sreg32(w, in.o[2]) & 63u) = sreg32(w, in.o[1]); }
static void x_readfirstlane(Machine &, Wave &w, const Inst &in) {
  const unsigned lane = w.exec ? (unsigned)__builtin_ctzll(w.exec) : 0u;
  swrite32(w, in.o[0], src32(w, in.o[1], lane));
}

// ---- scalar ops ---------------------------------------------------------------------------------------------------------
enum {
  S_MOV32, S_MOV64, S_MOVK, S_ADD_I32, S_ADDK, S_SUB_I32, S_MUL_I32, S_MUL_HI_U32, S_AND32, S_AND64, S_OR32, S_OR64, S_XOR32, S_XOR64, S_ANDN2_64, S_ORN2_64, S_ANDN2_32, S_NOT64,
  S_LSHL32, S_LSHL64, S_LSHR32, S_LSHR64, S_ASHR32, S_BFM32, S_BREV32, S_BCNT1_64, S_BCNT1_32, S_MIN_U32, S_MAX_U32, S_MIN_I32, S_MAX_I32, S_CSELECT32, S_CSELECT64, S_AND_SAVEEXEC, S_OR_SAVEEXEC,
  S_ANDN2_SAVEEXEC, S_GETREG, S_FF1_64, S_FF1_32, S_ADD_U32, S_SUB_U32, S_ADDC_U32, S_SUBB_U32, S_BFE_U32, S_ABS_I32, S_SEXT_I32_I16, S_NOT32, S_XNOR64, S_NAND64, S_NOR64, S_MUL_HI_I32, S_MEMTIME, S_BFE_I64, S_BFE_I32, S_MULK, S_BFE_U64, S_GETPC, S_FLBIT32, S_FLBIT64
};
static void x_salu(Machine &M, Wave &w, const Inst &in) {
  const Operand *o = in.o;
  switch (in.sub) {
  case S_MOV32: swrite32(w, o[0], sreg32(w, o[1])); break;
  case S_MOV64: swrite64(w, o[0], sreg64(w, o[1])); break;
  case S_MOVK: swrite32(w, o[0], (uint32_t)(int32_t)(int16_t)o[1].imm); break;
  case S_ADD_I32: {
    const int64_t a = (int32_t)sreg32(w, o[1]), b = (int32_t)sreg32(w, o[2]), r = a + b;
    w.scc = r > INT32_MAX || r < INT32_MIN;
    swrite32(w, o[0], (uint32_t)r);
    break;
  }
  case S_ADD_U32: {
    const uint64_t r = (uint64_t)sreg32(w, o[1]) + sreg32(w, o[2]);
    w.scc = r >> 32;
    swrite32(w, o[0], (uint32_t)r);
    break;
  }
  case S_ADDC_U32: {
    const uint64_t r = (uint64_t)sreg32(w, o[1]) + sreg32(w, o[2]) + (w.scc ? 1 : 0);
    w.scc = r >> 32;
    swrite32(w, o[0], (uint32_t)r);
    break;
  }
  case S_SUB_U32: {
    const uint32_t a = sreg32(w, o[1]), b = sreg32(w, o[2]);
    w.scc = b > a;
    swrite32(w, o[0], a - b);
    break;
  }
  case S_SUBB_U32: {
    const uint64_t a = sreg32(w, o[1]), b = (uint64_t)sreg32(w, o[2]) + (w.scc ? 1 : 0);
    w.scc = b > a;
    swrite32(w, o[0], (uint32_t)(a - b));
    break;
  }
  case S_ADDK: {
    const int64_t a = (int32_t)sreg32(w, o[0]), b = (int16_t)o[1].imm, r = a + b;
    w.scc = r > INT32_MAX || r < INT32_MIN;
    swrite32(w, o[0], (uint32_t)r);
    break;
  }
  case S_SUB_I32: {
    const int64_t a = (int32_t)sreg32(w, o[1]), b = (int32_t)sreg32(w, o[2]), r = a - b;
    w.scc = r > INT32_MAX || r < INT32_MIN;
    swrite32(w, o[0], (uint32_t)r);
    break;
  }
  case S_MUL_I32: swrite32(w, o[0], (uint32_t)((int32_t)sreg32(w, o[1]) * (int64_t)(int32_t)sreg32(w, o[2]))); break;
  case S_MUL_HI_I32: swrite32(w, o[0], (uint32_t)(((int64_t)(int32_t)sreg32(w, o[1]) * (int64_t)(int32_t)sreg32(w, o[2])) >> 32)); break;
  case S_MEMTIME: { static std::atomic<uint64_t> tick{0}; swrite64(w, o[0], tick.fetch_add(100)); break; }
  case S_MUL_HI_U32: swrite32(w, o[0], (uint32_t)(((uint64_t)sreg32(w, o[1]) * sreg32(w, o[2])) >> 32)); break;
#define SBIT32(expr) { const uint32_t a = sreg32(w, o[1]), b = sreg32(w, o[2]); const uint32_t r = (expr); w.scc = r != 0; swrite32(w, o[0], r); break; }
#define SBIT64(expr) { const uint64_t a = sreg64(w, o[1]), b = sreg64(w, o[2]); const uint64_t r = (expr); w.scc = r != 0; swrite64(w, o[0], r); break; }
  case S_AND32: SBIT32(a & b)
  case S_OR32: SBIT32(a | b)
  case S_XOR32: SBIT32(a ^ b)
  case S_ANDN2_32: SBIT32(a & ~b)
  case S_AND64: SBIT64(a & b)
  case S_OR64: SBIT64(a | b)
  case S_XOR64: SBIT64(a ^ b)
  case S_ANDN2_64: SBIT64(a & ~b)
  case S_ORN2_64: SBIT64(a | ~b)
  case S_XNOR64: SBIT64(~(a ^ b))
  case S_NAND64: SBIT64(~(a & b))
  case S_NOR64: SBIT64(~(a | b))
  case S_NOT64: { const uint64_t r = ~sreg64(w, o[1]); w.scc = r != 0; swrite64(w, o[0], r); break; }
  case S_NOT32: { const uint32_t r = ~sreg32(w, o[1]); w.scc = r != 0; swrite32(w, o[0], r); break; }
  case S_LSHL32: { const uint32_t r = sreg32(w, o[1]) << (sreg32(w, o[2]) & 31u); w.scc = r != 0; swrite32(w, o[0], r); break; }
  case S_LSHR32: { const uint32_t r = sreg32(w, o[1]) >> (sreg32(w, o[2]) & 31u); w.scc = r != 0; swrite32(w, o[0], r); break; }
  case S_ASHR32: { const uint32_t r = (uint32_t)((int32_t)sreg32(w, o[1]) >> (sreg32(w, o[2]) & 31u)); w.scc = r != 0; swrite32(w, o[0], r); break; }
  case S_LSHL64: { const uint64_t r = sreg64(w, o[1]) << (sreg32(w, o[2]) & 63u); w.scc = r != 0; swrite64(w, o[0], r); break; }
  case S_LSHR64: { const uint64_t r = sreg64(w, o[1]) >> (sreg32(w, o[2]) & 63u); w.scc = r != 0; swrite64(w, o[0], r); break; }
  case S_BFM32: swrite32(w, o[0], (uint32_t)(((1ull << (sreg32(w, o[1]) & 31u)) - 1ull) << (sreg32(w, o[2]) & 31u))); break;
  case S_BFE_U32: {
    const uint32_t a = sreg32(w, o[1]), b = sreg32(w, o[2]), off = b & 31u, wd = (b >> 16) & 0x7Fu;
    const uint32_t r = wd == 0 ? 0u : (wd >= 32 ? (a >> off) : ((a >> off) & ((1u << wd) - 1u)));
    w.scc = r != 0;
    swrite32(w, o[0], r);
    break;
  }
  case S_BFE_I64: case S_BFE_U64: {
    const uint64_t a = sreg64(w, o[1]);
    const uint32_t b = sreg32(w, o[2]), off = b & 63u, wd = (b >> 16) & 0x7Fu;
    uint64_t r = wd == 0 ? 0ull : (wd >= 64 ? (a >> off) : ((a >> off) & ((1ull << wd) - 1ull)));
    if (in.sub == S_BFE_I64 && wd > 0 && wd < 64 && ((r >> (wd - 1)) & 1)) r |= ~((1ull << wd) - 1ull);
    w.scc = r != 0;
    swrite64(w, o[0], r);
    break;
  }
  case S_BFE_I32: {
    const uint32_t a = sreg32(w, o[1]), b = sreg32(w, o[2]), off = b & 31u, wd = (b >> 16) & 0x7Fu;
    uint32_t r = wd == 0 ? 0u : (wd >= 32 ? (a >> off) : ((a >> off) & ((1u << wd) - 1u)));
    if (wd > 0 && wd < 32 && ((r >> (wd - 1)) & 1)) r |= ~((1u << wd) - 1u);
    w.scc = r != 0;
    swrite32(w, o[0], r);
    break;
  }
  case S_FLBIT32: { const uint32_t a = sreg32(w, o[1]); swrite32(w, o[0], a ? (uint32_t)__builtin_clz(a) : 0xFFFFFFFFu); break; }
  case S_FLBIT64: { const uint64_t a = sreg64(w, o[1]); swrite32(w, o[0], a ? (uint32_t)__builtin_clzll(a) : 0xFFFFFFFFu); break; }
  case S_GETPC: swrite64(w, o[0], 0ull); break; // (relocated symbol operands carry absolute addresses: see parse_operand)
  case S_MULK: swrite32(w, o[0], (uint32_t)((int32_t)sreg32(w, o[0]) * (int32_t)(int16_t)o[1].imm)); break;
  case S_BREV32: {
    uint32_t x = sreg32(w, o[1]), r = 0;
    for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i);
    swrite32(w, o[0], r);
    break;
  }
  case S_BCNT1_64: { const uint32_t r = (uint32_t)__builtin_popcountll(sreg64(w, o[1])); w.scc = r != 0; swrite32(w, o[0], r); break; }
  case S_BCNT1_32: { const uint32_t r = (uint32_t)__builtin_popcount(sreg32(w, o[1])); w.scc = r != 0; swrite32(w, o[0], r); break; }
  case S_FF1_64: { const uint64_t a = sreg64(w, o[1]); swrite32(w, o[0], a ? (uint32_t)__builtin_ctzll(a) : 0xFFFFFFFFu); break; }
  case S_FF1_32: { const uint32_t a = sreg32(w, o[1]); swrite32(w, o[0], a ? (uint32_t)__builtin_ctz(a) : 0xFFFFFFFFu); break; }
  case S_MIN_U32: { const uint32_t a = sreg32(w, o[1]), b = sreg32(w, o[2]); w.scc = a < b; swrite32(w, o[0], a < b ? a : b); break; }
  case S_MAX_U32: { const uint32_t a = sreg32(w, o[1]), b = sreg32(w, o[2]); w.scc = a > b; swrite32(w, o[0], a > b ? a : b); break; }
  case S_MIN_I32: { const int32_t a = (int32_t)sreg32(w, o[1]), b = (int32_t)sreg32(w, o[2]); w.scc = a < b; swrite32(w, o[0], (uint32_t)(a < b ? a : b)); break; }
  case S_MAX_I32: { const int32_t a = (int32_t)sreg32(w, o[1]), b = (int32_t)sreg32(w, o[2]); w.scc = a > b; swrite32(w, o[0], (uint32_t)(a > b ? a : b)); break; }
  case S_ABS_I32: { const int32_t a = (int32_t)sreg32(w, o[1]); const uint32_t r = (uint32_t)(a < 0 ? -a : a); w.scc = r != 0; swrite32(w, o[0], r); break; }
  case S_SEXT_I32_I16: swrite32(w, o[0], (uint32_t)(int32_t)(int16_t)sreg32(w, o[1])); break;
  case S_CSELECT32: swrite32(w, o[0], w.scc ? sreg32(w, o[1]) : sreg32(w, o[2])); break;
  case S_CSELECT64: swrite64(w, o[0], w.scc ? sreg64(w, o[1]) : sreg64(w, o[2])); break;
  case S_AND_SAVEEXEC: { const uint64_t old = w.exec, src = sreg64(w, o[1]); w.exec = src & old; swrite64(w, o[0], old); w.scc = w.exec != 0; break; }
  case S_OR_SAVEEXEC: { const uint64_t old = w.exec, src = sreg64(w, o[1]); w.exec = src | old; swrite64(w, o[0], old); w.scc = w.exec != 0; break; }
  case S_ANDN2_SAVEEXEC: { const uint64_t old = w.exec, src = sreg64(w, o[1]); w.exec = src & ~old; swrite64(w, o[0], old); w.scc = w.exec != 0; break; }
  case S_GETREG: swrite32(w, o[0], M.block_id & 7u); break; // hwreg(HW_REG_XCC_ID): which XCD -- as the C++ emulator answers
  }
}
enum { SC_EQ, SC_LG, SC_GT, SC_GE, SC_LT, SC_LE, SC_BITCMP0, SC_BITCMP1 };
static void x_scmp(Machine &, Wave &w, const Inst &in) {
  const int ty = in.sub >> 8, p = in.sub & 0xFF;
  bool r = false;
  if (p == SC_BITCMP0 || p == SC_BITCMP1) {
    const bool bit = ty == T_U64 ? ((sreg64(w, in.o[0]) >> (sreg32(w, in.o[1]) & 63u)) & 1) : ((sreg32(w, in.o[0]) >> (sreg32(w, in.o[1]) & 31u)) & 1);
    w.scc = p == SC_BITCMP1 ? bit : !bit;
    return;
  }
  if (ty == T_U64) {
    const uint64_t a = sreg64(w, in.o[0]), b = sreg64(w, in.o[1]);
    r = p == SC_EQ ? a == b : a != b;
  } else if (ty == T_I32) {
    const int32_t a = (int32_t)sreg32(w, in.o[0]), b = (int32_t)sreg32(w, in.o[1]);
    r = p == SC_EQ ? a == b : p == SC_LG ? a != b : p == SC_GT ? a > b : p == SC_GE ? a >= b : p == SC_LT ? a < b : a <= b;
  } else {
    const uint32_t a = sreg32(w, in.o[0]), b = sreg32(w, in.o[1]);
    r = p == SC_EQ ? a == b : p == SC_LG ? a != b : p == SC_GT ? a > b : p == SC_GE ? a >= b : p == SC_LT ? a < b : a <= b;
  }
  w.scc = r;
}
enum { B_ALWAYS, B_SCC0, B_SCC1, B_VCCZ, B_VCCNZ, B_EXECZ, B_EXECNZ };
static void x_branch(Machine &, Wave &w, const Inst &in) {
  bool take = false;
  switch (in.sub) {
  case B_ALWAYS: take = true; break;
  case B_SCC0: take = !w.scc; break;
  case B_SCC1: take = w.scc; break;
  case B_VCCZ: take = w.vcc == 0; break;
  case B_VCCNZ: take = w.vcc != 0; break;
  case B_EXECZ: take = w.exec == 0; break;
  case B_EXECNZ: take = w.exec != 0; break;
  }
  if (take) w.pc = (size_t)in.o[0].label - 1; // (the loop adds 1)
}
static void x_nop(Machine &, Wave &, const Inst &) {}
// opcodes that sit on paths the kernels' inputs never take (Payne-Hanek reduction of huge sincos arguments): loading the kernel is fine, executing one aborts
static void x_unmodelled(Machine &, Wave &, const Inst &in) { die(in, "instruction not modelled (an input took a path the interpreter does not cover)"); }
static void x_sleep(Machine &M, Wave &, const Inst &) { M.yield = true; }
static void x_barrier(Machine &M, Wave &w, const Inst &) { w.at_barrier = true; M.yield = true; }
static void x_endpgm(Machine &M, Wave &w, const Inst &) { w.done = true; M.yield = true; }

// ---- SMEM: s_load_dword{,x2,x4,x8,x16} sdst, sbase, offset --------------------------------------------------------------
static void x_sload(Machine &, Wave &w, const Inst &in) {
  const uint64_t base = sreg64(w, in.o[1]);
  const uint64_t off = in.o[2].kind == K_IMM ? in.o[2].imm : (uint64_t)sreg32(w, in.o[2]);
  const uint8_t *p = reinterpret_cast<const uint8_t *>(base + off + (uint64_t)(int64_t)in.offset);
  if (base + off < 4096) die(in, "s_load from a null page address");
  for (unsigned i = 0; i < in.sub; ++i) {
    uint32_t v;
    memcpy(&v, p + 4 * i, 4);
    swrite32(w, in.o[0], v, i);
  }
}

// ---- LDS ------------------------------------------------------------------------------------------------------------------
enum { L_READ, L_WRITE, L_READ2, L_WRITE2, L_READ2ST64, L_WRITE2ST64, L_READ_U8, L_WRITE_B8, L_ADD_RTN, L_ADD, L_CMPST_RTN, L_WRXCHG_RTN, L_BPERMUTE, L_READ_U16, L_WRITE_B16, L_MAX_RTN_U32, L_MIN_RTN_U32, L_OR_RTN, L_AND_RTN, L_MAX_U32, L_MIN_U32, L_OR, L_AND, L_MIN_I32, L_MAX_I32 };
// sub = kind | words << 8
static void x_lds(Machine &M, Wave &w, const Inst &in) {
  const int kind = in.sub & 0xFF;
  const unsigned words = in.sub >> 8;
  if (kind == L_BPERMUTE) { // ds_bpermute_b32 vdst, addr, data: lane i gets data of lane (addr[i] / 4) % 64, 0 when that lane is off
    uint32_t tmp[64];
    for (unsigned l = 0; l < 64; ++l) {
      const unsigned srcl = ((w.vr(in.o[1].reg, l) + (uint32_t)in.offset) >> 2) & 63u;
      tmp[l] = ((w.exec >> srcl) & 1) ? w.vr(in.o[2].reg, srcl) : 0u;
    }
    FOR_LANES(w) w.vr(in.o[0].reg, lane) = tmp[lane];
    return;
  }
  FOR_LANES(w) {
    switch (kind) {
    case L_READ: {
      const uint32_t a = w.vr(in.o[1].reg, lane) + (uint32_t)in.offset;
      const uint8_t *p = lds_ptr(M, a, 4 * words, in);
      for (unsigned i = 0; i < words; ++i) memcpy(&w.vr(in.o[0].reg + i, lane), p + 4 * i, 4);
      break;
    }
    case L_READ_U8: w.vr(in.o[0].reg, lane) = *lds_ptr(M, w.vr(in.o[1].reg, lane) + (uint32_t)in.offset, 1, in); break;
    case L_READ_U16: { uint16_t t; memcpy(&t, lds_ptr(M, w.vr(in.o[1].reg, lane) + (uint32_t)in.offset, 2, in), 2); w.vr(in.o[0].reg, lane) = t; break; }
    case L_WRITE: {
      const uint32_t a = w.vr(in.o[0].reg, lane) + (uint32_t)in.offset;
      uint8_t *p = lds_ptr(M, a, 4 * words, in);
      for (unsigned i = 0; i < words; ++i) { const uint32_t t = src32(w, in.o[1], lane, i); memcpy(p + 4 * i, &t, 4); }
      break;
    }
    case L_WRITE_B8: *lds_ptr(M, w.vr(in.o[0].reg, lane) + (uint32_t)in.offset, 1, in) = (uint8_t)src32(w, in.o[1], lane); break;
    case L_WRITE_B16: { const uint16_t t = (uint16_t)src32(w, in.o[1], lane); memcpy(lds_ptr(M, w.vr(in.o[0].reg, lane) + (uint32_t)in.offset, 2, in), &t, 2); break; }
    case L_READ2: case L_READ2ST64: { // words = dwords per element (1: b32, 2: b64)
      const uint32_t scale = 4u * words * (kind == L_READ2ST64 ? 64u : 1u), a = w.vr(in.o[1].reg, lane);
      const uint8_t *p0 = lds_ptr(M, a + (uint32_t)in.offset0 * scale, 4 * words, in), *p1 = lds_ptr(M, a + (uint32_t)in.offset1 * scale, 4 * words, in);
      for (unsigned i = 0; i < words; ++i) memcpy(&w.vr(in.o[0].reg + i, lane), p0 + 4 * i, 4);
      for (unsigned i = 0; i < words; ++i) memcpy(&w.vr(in.o[0].reg + words + i, lane), p1 + 4 * i, 4);
      break;
    }
    case L_WRITE2: case L_WRITE2ST64: {
      const uint32_t scale = 4u * words * (kind == L_WRITE2ST64 ? 64u : 1u), a = w.vr(in.o[0].reg, lane);
      uint8_t *p0 = lds_ptr(M, a + (uint32_t)in.offset0 * scale, 4 * words, in), *p1 = lds_ptr(M, a + (uint32_t)in.offset1 * scale, 4 * words, in);
      for (unsigned i = 0; i < words; ++i) { const uint32_t t = src32(w, in.o[1], lane, i); memcpy(p0 + 4 * i, &t, 4); }
      for (unsigned i = 0; i < words; ++i) { const uint32_t t = src32(w, in.o[2], lane, i); memcpy(p1 + 4 * i, &t, 4); }
      break;
    }
    case L_ADD_RTN: case L_ADD: { // ds_add_rtn_u32 vdst, addr, data ; ds_add_u32 addr, data
      const bool rtn = kind == L_ADD_RTN;
      const Operand &addr = in.o[rtn ? 1 : 0], &data = in.o[rtn ? 2 : 1];
      uint8_t *p = lds_ptr(M, w.vr(addr.reg, lane) + (uint32_t)in.offset, 4 * words, in);
      if (words == 1) {
        uint32_t old; memcpy(&old, p, 4);
        const uint32_t nw = old + src32(w, data, lane);
        memcpy(p, &nw, 4);
        if (rtn) w.vr(in.o[0].reg, lane) = old;
      } else {
        uint64_t old; memcpy(&old, p, 8);
        const uint64_t nw = old + src64(w, data, lane);
        memcpy(p, &nw, 8);
        if (rtn) dst64(w, in.o[0], lane, old);
      }
      break;
    }
    case L_MAX_RTN_U32: case L_MIN_RTN_U32: case L_OR_RTN: case L_AND_RTN: {
      uint8_t *p = lds_ptr(M, w.vr(in.o[1].reg, lane) + (uint32_t)in.offset, 4, in);
      uint32_t old; memcpy(&old, p, 4);
      const uint32_t d = src32(w, in.o[2], lane);
      const uint32_t nw = kind == L_MAX_RTN_U32 ? std::max(old, d) : kind == L_MIN_RTN_U32 ? std::min(old, d) : kind == L_OR_RTN ? (old | d) : (old & d);
      memcpy(p, &nw, 4);
      w.vr(in.o[0].reg, lane) = old;
      break;
    }
    case L_MIN_I32: case L_MAX_I32: {
      uint8_t *p = lds_ptr(M, w.vr(in.o[0].reg, lane) + (uint32_t)in.offset, 4, in);
      int32_t old; memcpy(&old, p, 4);
      const int32_t d = (int32_t)src32(w, in.o[1], lane), nw = kind == L_MIN_I32 ? std::min(old, d) : std::max(old, d);
      memcpy(p, &nw, 4);
      break;
    }
    case L_MAX_U32: case L_MIN_U32: case L_OR: case L_AND: { // addr, data
      uint8_t *p = lds_ptr(M, w.vr(in.o[0].reg, lane) + (uint32_t)in.offset, 4, in);
      uint32_t old; memcpy(&old, p, 4);
      const uint32_t d = src32(w, in.o[1], lane);
      const uint32_t nw = kind == L_MAX_U32 ? std::max(old, d) : kind == L_MIN_U32 ? std::min(old, d) : kind == L_OR ? (old | d) : (old & d);
      memcpy(p, &nw, 4);
      break;
    }
    case L_CMPST_RTN: { // vdst, addr, cmp, new
      uint8_t *p = lds_ptr(M, w.vr(in.o[1].reg, lane) + (uint32_t)in.offset, 4, in);
      uint32_t old; memcpy(&old, p, 4);
      if (old == src32(w, in.o[2], lane)) { const uint32_t nw = src32(w, in.o[3], lane); memcpy(p, &nw, 4); }
      w.vr(in.o[0].reg, lane) = old;
      break;
    }
    case L_WRXCHG_RTN: { // vdst, addr, data
      uint8_t *p = lds_ptr(M, w.vr(in.o[1].reg, lane) + (uint32_t)in.offset, 4 * words, in);
      for (unsigned i = 0; i < words; ++i) {
        uint32_t old; memcpy(&old, p + 4 * i, 4);
        const uint32_t nw = src32(w, in.o[2], lane, i);
        memcpy(p + 4 * i, &nw, 4);
        w.vr(in.o[0].reg + i, lane) = old;
      }
      break;
    }
    }
  }
}

// ---- flat / global / scratch ----------------------------------------------------------------------------------------------
// address of a flat / global access: vaddr is a 64-bit VGPR pair (saddr "off") or a 32-bit VGPR offset added to an SGPR pair
static inline uint64_t vm_addr(Wave &w, const Operand &vaddr, const Operand *saddr, unsigned lane, int32_t offset) {
  uint64_t a;
  if (saddr && saddr->kind != K_OFF && saddr->kind != K_NONE) a = sreg64(w, *saddr) + (uint64_t)w.vr(vaddr.reg, lane);
  else a = (uint64_t)w.vr(vaddr.reg, lane) | ((uint64_t)w.vr(vaddr.reg + 1, lane) << 32);
  return a + (uint64_t)(int64_t)offset;
}
enum { VM_LOAD, VM_STORE, VM_ATOMIC_ADD, VM_LOAD_U8, VM_STORE_B8, VM_LOAD_U16, VM_STORE_B16, VM_ATOMIC_UMAX, VM_ATOMIC_CMPSWAP, VM_ATOMIC_SWAP, VM_ATOMIC_UMIN, VM_ATOMIC_OR, VM_ATOMIC_AND, VM_LOAD_I8, VM_LOAD_I16 };
// sub = kind | words << 8 | is_global << 15
static void x_vmem(Machine &M, Wave &w, const Inst &in) {
  const int kind = in.sub & 0xFF;
  const unsigned words = (in.sub >> 8) & 0x7F;
  const bool global = (in.sub >> 15) & 1;
  FOR_LANES(w) {
    switch (kind) {
    case VM_LOAD: case VM_LOAD_U8: case VM_LOAD_U16: case VM_LOAD_I8: case VM_LOAD_I16: { // [global] vdst, vaddr, saddr|off ; [flat] vdst, vaddr
      const uint64_t a = vm_addr(w, in.o[1], global ? &in.o[2] : nullptr, lane, in.offset);
      const size_t bytes = kind == VM_LOAD ? 4 * words : ((kind == VM_LOAD_U8 || kind == VM_LOAD_I8) ? 1 : 2);
      const uint8_t *p = flat_ptr(M, w, lane, a, bytes, in);
      if (kind == VM_LOAD) for (unsigned i = 0; i < words; ++i) memcpy(&w.vr(in.o[0].reg + i, lane), p + 4 * i, 4);
      else if (kind == VM_LOAD_U8) w.vr(in.o[0].reg, lane) = *p;
      else if (kind == VM_LOAD_I8) w.vr(in.o[0].reg, lane) = (uint32_t)(int32_t)(int8_t)*p;
      else { uint16_t t; memcpy(&t, p, 2); w.vr(in.o[0].reg, lane) = kind == VM_LOAD_I16 ? (uint32_t)(int32_t)(int16_t)t : (uint32_t)t; }
      break;
    }
    case VM_STORE: case VM_STORE_B8: case VM_STORE_B16: { // [global] vaddr, vdata, saddr|off ; [flat] vaddr, vdata
      const uint64_t a = vm_addr(w, in.o[0], global ? &in.o[2] : nullptr, lane, in.offset);
      const size_t bytes = kind == VM_STORE ? 4 * words : (kind == VM_STORE_B8 ? 1 : 2);
      uint8_t *p = flat_ptr(M, w, lane, a, bytes, in);
      if (kind == VM_STORE) for (unsigned i = 0; i < words; ++i) { const uint32_t t = src32(w, in.o[1], lane, i); memcpy(p + 4 * i, &t, 4); }
      else if (kind == VM_STORE_B8) *p = (uint8_t)src32(w, in.o[1], lane);
      else { const uint16_t t = (uint16_t)src32(w, in.o[1], lane); memcpy(p, &t, 2); }
      break;
    }
    default: { // atomics: with a return value (sc0): vdst, vaddr, vdata [, saddr] ; without: vaddr, vdata [, saddr]
      const bool rtn = in.sc0;
      const Operand &va = in.o[rtn ? 1 : 0], &vd = in.o[rtn ? 2 : 1];
      const Operand *sa = global ? &in.o[rtn ? 3 : 2] : nullptr;
      const uint64_t a = vm_addr(w, va, sa, lane, in.offset);
      uint8_t *p = flat_ptr(M, w, lane, a, 4 * words, in);
      if (words == 1) {
        uint32_t *q = reinterpret_cast<uint32_t *>(p);
        const uint32_t d = src32(w, vd, lane);
        uint32_t old = 0;
        switch (kind) {
        case VM_ATOMIC_ADD: old = __atomic_fetch_add(q, d, __ATOMIC_SEQ_CST); break;
        case VM_ATOMIC_UMAX: { old = __atomic_load_n(q, __ATOMIC_SEQ_CST); while (old < d && !__atomic_compare_exchange_n(q, &old, d, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} break; }
        case VM_ATOMIC_UMIN: { old = __atomic_load_n(q, __ATOMIC_SEQ_CST); while (old > d && !__atomic_compare_exchange_n(q, &old, d, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} break; }
        case VM_ATOMIC_OR: old = __atomic_fetch_or(q, d, __ATOMIC_SEQ_CST); break;
        case VM_ATOMIC_AND: old = __atomic_fetch_and(q, d, __ATOMIC_SEQ_CST); break;
        case VM_ATOMIC_SWAP: old = __atomic_exchange_n(q, d, __ATOMIC_SEQ_CST); break;
        case VM_ATOMIC_CMPSWAP: { // vdata = {new, cmp}
          old = src32(w, vd, lane, 1);
          __atomic_compare_exchange_n(q, &old, d, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
          break;
        }
        }
        if (rtn) w.vr(in.o[0].reg, lane) = old;
      } else {
        uint64_t *q = reinterpret_cast<uint64_t *>(p);
        const uint64_t d = src64(w, vd, lane);
        uint64_t old = 0;
        switch (kind) {
        case VM_ATOMIC_ADD: old = __atomic_fetch_add(q, d, __ATOMIC_SEQ_CST); break;
        case VM_ATOMIC_UMAX: { old = __atomic_load_n(q, __ATOMIC_SEQ_CST); while (old < d && !__atomic_compare_exchange_n(q, &old, d, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} break; }
        case VM_ATOMIC_SWAP: old = __atomic_exchange_n(q, d, __ATOMIC_SEQ_CST); break;
        default: die(in, "64-bit atomic not implemented");
        }
        if (rtn) dst64(w, in.o[0], lane, old);
      }
      break;
    }
    }
  }
}
// scratch_load_dword vdst, vaddr|off, saddr|off offset ; scratch_store_dword vaddr|off, vdata, saddr|off offset
static void x_scratch(Machine &, Wave &w, const Inst &in) {
  const bool load = (in.sub & 0xFF) == VM_LOAD;
  const unsigned words = in.sub >> 8;
  const Operand &va = in.o[load ? 1 : 0], &sa = in.o[2];
  FOR_LANES(w) {
    uint32_t a = (uint32_t)in.offset;
    if (va.kind == K_VGPR) a += w.vr(va.reg, lane);
    if (sa.kind == K_SGPR) a += w.s[sa.reg];
    if ((size_t)a + 4 * words > w.scratch_stride) die(in, "scratch access beyond the private segment");
    uint8_t *p = w.scratch.data() + (size_t)lane * w.scratch_stride + a;
    if (load) for (unsigned i = 0; i < words; ++i) memcpy(&w.vr(in.o[0].reg + i, lane), p + 4 * i, 4);
    else for (unsigned i = 0; i < words; ++i) { const uint32_t t = src32(w, in.o[1], lane, i); memcpy(p + 4 * i, &t, 4); }
  }
}

// ---------------------------------------------------------------------------------------------------------------- parsing
struct OpDef {
  ExecFn fn;
  Cls cls;
  uint16_t sub;
  const char *sig; // one char per operand: the type a CONSTANT in that position has -- d f64, f f32, i/u/b 32-bit integer, U 64-bit integer, - none
};
static std::unordered_map<std::string, OpDef> &optable() {
  static std::unordered_map<std::string, OpDef> t;
  if (!t.empty()) return t;
  auto add = [&](const char *names, ExecFn fn, Cls c, uint16_t sub, const char *sig) {
    std::stringstream ss(names);
    std::string n;
    while (ss >> n) t[n] = OpDef{fn, c, sub, sig};
  };
  // ---- f64
  add("v_add_f64", x_f64, C_VALU, D_ADD, "ddd");
  add("v_mul_f64", x_f64, C_VALU, D_MUL, "ddd");
  add("v_fma_f64", x_f64, C_VALU, D_FMA, "dddd");
  add("v_fmac_f64_e32 v_fmac_f64_e64", x_f64, C_VALU, D_FMAC, "ddd");
  add("v_min_f64", x_f64, C_VALU, D_MIN, "ddd");
  add("v_max_f64", x_f64, C_VALU, D_MAX, "ddd");
  add("v_ldexp_f64", x_f64, C_VALU, D_LDEXP, "ddi");
  add("v_rcp_f64_e32 v_rcp_f64_e64", x_f64, C_VALU, D_RCP, "dd");
  add("v_rsq_f64_e32 v_rsq_f64_e64", x_f64, C_VALU, D_RSQ, "dd");
  add("v_fract_f64_e32 v_fract_f64_e64", x_f64, C_VALU, D_FRACT, "dd");
  add("v_rndne_f64_e32 v_rndne_f64_e64", x_f64, C_VALU, D_RNDNE, "dd");
  add("v_div_scale_f64", x_div_scale_f64, C_VALU, 0, "dUddd");
  add("v_div_fmas_f64", x_div_fmas_f64, C_VALU, 0, "dddd");
  add("v_div_fixup_f64", x_div_fixup_f64, C_VALU, 0, "dddd");
  add("v_cmp_class_f64_e32 v_cmp_class_f64_e64", x_vcmp_class_f64, C_VALU, 0, "Udu");
  add("v_cmp_class_f32_e32 v_cmp_class_f32_e64", x_vcmp_class_f32, C_VALU, 0, "Ufu");
  // ---- f32
  add("v_add_f32_e32 v_add_f32_e64", x_f32, C_VALU, F_ADD, "fff");
  add("v_sub_f32_e32 v_sub_f32_e64", x_f32, C_VALU, F_SUB, "fff");
  add("v_mul_f32_e32 v_mul_f32_e64", x_f32, C_VALU, F_MUL, "fff");
  add("v_fma_f32", x_f32, C_VALU, F_FMA, "ffff");
  add("v_fmac_f32_e32 v_fmac_f32_e64", x_f32, C_VALU, F_FMAC, "fff");
  add("v_min_f32_e32 v_min_f32_e64", x_f32, C_VALU, F_MIN, "fff");
  add("v_max_f32_e32 v_max_f32_e64", x_f32, C_VALU, F_MAX, "fff");
  add("v_min3_f32", x_f32, C_VALU, F_MIN3, "ffff");
  add("v_max3_f32", x_f32, C_VALU, F_MAX3, "ffff");
  add("v_rcp_f32_e32 v_rcp_f32_e64 v_rcp_iflag_f32_e32 v_rcp_iflag_f32_e64", x_f32, C_VALU, F_RCP, "ff");
  add("v_pk_mul_f32", x_pk_f32, C_VALU, F_MUL, "fff");
  add("v_pk_add_f32", x_pk_f32, C_VALU, F_ADD, "fff");
  add("v_pk_fma_f32", x_pk_f32, C_VALU, F_FMA, "ffff");
  add("v_div_scale_f32", x_div_scale_f32, C_VALU, 0, "fUfff");
  add("v_div_fmas_f32", x_div_fmas_f32, C_VALU, 0, "ffff");
  add("v_div_fixup_f32", x_div_fixup_f32, C_VALU, 0, "ffff");
  // ---- conversions
  add("v_cvt_f32_f64_e32 v_cvt_f32_f64_e64", x_cvt, C_VALU, CV_F32_F64, "fd");
  add("v_cvt_f64_f32_e32 v_cvt_f64_f32_e64", x_cvt, C_VALU, CV_F64_F32, "df");
  add("v_cvt_f64_i32_e32 v_cvt_f64_i32_e64", x_cvt, C_VALU, CV_F64_I32, "di");
  add("v_cvt_f64_u32_e32 v_cvt_f64_u32_e64", x_cvt, C_VALU, CV_F64_U32, "du");
  add("v_cvt_i32_f64_e32 v_cvt_i32_f64_e64", x_cvt, C_VALU, CV_I32_F64, "id");
  add("v_cvt_u32_f64_e32 v_cvt_u32_f64_e64", x_cvt, C_VALU, CV_U32_F64, "ud");
  add("v_cvt_f32_i32_e32 v_cvt_f32_i32_e64", x_cvt, C_VALU, CV_F32_I32, "fi");
  add("v_cvt_f32_u32_e32 v_cvt_f32_u32_e64", x_cvt, C_VALU, CV_F32_U32, "fu");
  add("v_cvt_u32_f32_e32 v_cvt_u32_f32_e64", x_cvt, C_VALU, CV_U32_F32, "uf");
  add("v_cvt_i32_f32_e32 v_cvt_i32_f32_e64", x_cvt, C_VALU, CV_I32_F32, "if");
  // ---- integer
  add("v_mov_b32_e32 v_mov_b32_e64", x_int, C_VALU, I_MOV, "bb");
  add("v_mov_b64_e32 v_mov_b64_e64", x_int, C_VALU, I_MOV64, "UU");
  add("v_add_u32_e32 v_add_u32_e64", x_int, C_VALU, I_ADD, "uuu");
  add("v_sub_u32_e32 v_sub_u32_e64", x_int, C_VALU, I_SUB, "uuu");
  add("v_subrev_u32_e32 v_subrev_u32_e64", x_int, C_VALU, I_SUBREV, "uuu");
  add("v_and_b32_e32 v_and_b32_e64", x_int, C_VALU, I_AND, "bbb");
  add("v_or_b32_e32 v_or_b32_e64", x_int, C_VALU, I_OR, "bbb");
  add("v_xor_b32_e32 v_xor_b32_e64", x_int, C_VALU, I_XOR, "bbb");
  add("v_not_b32_e32 v_not_b32_e64", x_int, C_VALU, I_NOT, "bb");
  add("v_lshlrev_b32_e32 v_lshlrev_b32_e64", x_int, C_VALU, I_LSHLREV, "buu");
  add("v_lshrrev_b32_e32 v_lshrrev_b32_e64", x_int, C_VALU, I_LSHRREV, "buu");
  add("v_ashrrev_i32_e32 v_ashrrev_i32_e64", x_int, C_VALU, I_ASHRREV, "bui");
  add("v_lshlrev_b64", x_int, C_VALU, I_LSHLREV64, "UuU");
  add("v_lshrrev_b64", x_int, C_VALU, I_LSHRREV64, "UuU");
  add("v_add3_u32", x_int, C_VALU, I_ADD3, "uuuu");
  add("v_or3_b32", x_int, C_VALU, I_OR3, "bbbb");
  add("v_and_or_b32", x_int, C_VALU, I_AND_OR, "bbbb");
  add("v_lshl_add_u32", x_int, C_VALU, I_LSHL_ADD, "uuuu");
  add("v_add_lshl_u32", x_int, C_VALU, I_ADD_LSHL, "uuuu");
  add("v_lshl_or_b32", x_int, C_VALU, I_LSHL_OR, "uuuu");
  add("v_lshl_add_u64", x_int, C_VALU, I_LSHL_ADD64, "UUuU");
  add("v_bfe_u32", x_int, C_VALU, I_BFE_U32, "uuuu");
  add("v_bfrev_b32_e32 v_bfrev_b32_e64", x_int, C_VALU, I_BFREV, "bb");
  add("v_bcnt_u32_b32 v_bcnt_u32_b32_e64", x_int, C_VALU, I_BCNT, "uuu");
  add("v_mul_lo_u32", x_int, C_VALU, I_MUL_LO, "uuu");
  add("v_mul_hi_u32", x_int, C_VALU, I_MUL_HI, "uuu");
  add("v_mul_u32_u24_e32 v_mul_u32_u24_e64", x_int, C_VALU, I_MUL_U24, "uuu");
  add("v_mad_u32_u24", x_int, C_VALU, I_MAD_U24, "uuuu");
  add("v_min_i32_e32 v_min_i32_e64", x_int, C_VALU, I_MIN_I32, "iii");
  add("v_max_i32_e32 v_max_i32_e64", x_int, C_VALU, I_MAX_I32, "iii");
  add("v_min_u32_e32 v_min_u32_e64", x_int, C_VALU, I_MIN_U32, "uuu");
  add("v_max_u32_e32 v_max_u32_e64", x_int, C_VALU, I_MAX_U32, "uuu");
  add("v_mbcnt_lo_u32_b32", x_int, C_VALU, I_MBCNT_LO, "uuu");
  add("v_mbcnt_hi_u32_b32", x_int, C_VALU, I_MBCNT_HI, "uuu");
  add("v_bitop3_b32", x_int, C_VALU, I_BITOP3, "bbbb");
  add("v_cndmask_b32_e32 v_cndmask_b32_e64", x_int, C_VALU, I_CNDMASK, "bbbU");
  add("v_add_co_u32_e32 v_add_co_u32_e64", x_carry, C_VALU, CO_ADD, "uUuu");
  add("v_sub_co_u32_e32 v_sub_co_u32_e64", x_carry, C_VALU, CO_SUB, "uUuu");
  add("v_addc_co_u32_e32 v_addc_co_u32_e64", x_carry, C_VALU, CO_ADDC, "uUuuU");
  add("v_subb_co_u32_e32 v_subb_co_u32_e64", x_carry, C_VALU, CO_SUBB, "uUuuU");
  add("v_subbrev_co_u32_e32 v_subbrev_co_u32_e64", x_carry, C_VALU, CO_SUBBREV, "uUuuU");
  add("v_subrev_co_u32_e32 v_subrev_co_u32_e64", x_carry, C_VALU, CO_SUBREV, "uUuu");
  add("v_mad_u64_u32", x_mad64, C_VALU, 0, "UUuuU");
  add("v_mad_i64_i32", x_mad64, C_VALU, 1, "UUiiU");
  add("v_readfirstlane_b32", x_readfirstlane, C_VALU, 0, "bb");
  add("v_pk_mov_b32", x_pk_mov, C_VALU, 0, "UUU");
  add("v_add_u32_sdwa", x_sdwa, C_VALU, SD_ADD_U32, "uuu");
  add("v_min_u16_sdwa", x_sdwa, C_VALU, SD_MIN_U16, "uuu");
  add("v_or_b32_sdwa", x_sdwa, C_VALU, SD_OR_B32, "bbb");
  add("v_and_b32_sdwa", x_sdwa, C_VALU, SD_AND_B32, "bbb");
  add("v_mov_b32_sdwa", x_sdwa, C_VALU, SD_MOV_B32, "bb");
  add("v_lshlrev_b32_sdwa", x_sdwa, C_VALU, SD_LSHLREV_B32, "buu");
  add("v_cmp_gt_u16_sdwa", x_sdwa, C_VALU, SD_CMP_GT_U16, "Uuu");
  add("v_cmp_lt_u16_sdwa", x_sdwa, C_VALU, SD_CMP_LT_U16, "Uuu");
  add("v_cmp_ne_u16_sdwa", x_sdwa, C_VALU, SD_CMP_NE_U16, "Uuu");
  add("v_cmp_eq_u16_sdwa", x_sdwa, C_VALU, SD_CMP_EQ_U16, "Uuu");
  add("v_trunc_f32_e32 v_trunc_f32_e64", x_f32, C_VALU, F_TRUNC, "ff");
  add("v_rsq_f32_e32 v_rsq_f32_e64", x_f32, C_VALU, F_RSQ, "ff");
  add("v_trunc_f64_e32 v_trunc_f64_e64", x_f64, C_VALU, D_TRUNC, "dd");
  add("v_ceil_f64_e32 v_ceil_f64_e64", x_f64, C_VALU, D_CEIL, "dd");
  add("v_frexp_mant_f64_e32 v_frexp_mant_f64_e64", x_f64, C_VALU, D_FREXP_MANT, "dd");
  add("v_frexp_exp_i32_f64_e32 v_frexp_exp_i32_f64_e64", x_cvt, C_VALU, CV_FREXP_EXP_F64, "id");
  add("v_floor_f64_e32 v_floor_f64_e64", x_f64, C_VALU, D_FLOOR, "dd");
  add("v_add_u16_e32 v_add_u16_e64", x_int, C_VALU, I_ADD_U16, "uuu");
  add("v_floor_f32_e32 v_floor_f32_e64", x_f32, C_VALU, F_FLOOR, "ff");
  add("v_readlane_b32", x_readlane, C_VALU, 0, "bbu");
  add("v_writelane_b32", x_writelane, C_VALU, 0, "bbu");
  add("v_mul_hi_u32_u24_e32 v_mul_hi_u32_u24_e64", x_int, C_VALU, I_MUL_HI_U24, "uuu");
  add("v_sqrt_f32_e32 v_sqrt_f32_e64", x_f32, C_VALU, F_SQRT, "ff");
  add("v_alignbit_b32", x_int, C_VALU, I_ALIGNBIT, "bbbu");
  add("v_ffbl_b32_e32 v_ffbl_b32_e64", x_int, C_VALU, I_FFBL, "bb");
  add("v_ffbh_u32_e32 v_ffbh_u32_e64", x_int, C_VALU, I_FFBH, "bb");
  add("v_mad_u32_u16", x_int, C_VALU, I_MAD_U16, "uuuu");
  add("v_mul_i32_i24_e32 v_mul_i32_i24_e64", x_int, C_VALU, I_MUL_I24, "iii");
  add("v_mad_i32_i24", x_int, C_VALU, I_MAD_I24, "iiii");
  add("v_xad_u32", x_int, C_VALU, I_XAD, "uuuu");
  add("v_sub_co_u32_e32 v_sub_co_u32_e64", x_carry, C_VALU, CO_SUB, "uUuu");
  // ---- compares
  struct { const char *n; int p; } fp[] = {{"lt", P_LT}, {"eq", P_EQ}, {"le", P_LE}, {"gt", P_GT}, {"lg", P_LG}, {"ge", P_GE}, {"o", P_O}, {"u", P_U}, {"nge", P_NGE},
                                          {"nlg", P_NLG}, {"ngt", P_NGT}, {"nle", P_NLE}, {"neq", P_NEQ}, {"nlt", P_NLT}};
  for (auto &e : fp)
    for (const char *suf : {"_e32", "_e64"}) {
      t[std::string("v_cmp_") + e.n + "_f64" + suf] = OpDef{x_vcmp, C_VALU, (uint16_t)((T_F64 << 8) | e.p), "Udd"};
      t[std::string("v_cmp_") + e.n + "_f32" + suf] = OpDef{x_vcmp, C_VALU, (uint16_t)((T_F32 << 8) | e.p), "Uff"};
    }
  struct { const char *n; int p; } ip[] = {{"lt", P_LT}, {"eq", P_EQ}, {"le", P_LE}, {"gt", P_GT}, {"ne", P_NE}, {"ge", P_GE}};
  for (auto &e : ip)
    for (const char *suf : {"_e32", "_e64"}) {
      t[std::string("v_cmp_") + e.n + "_i32" + suf] = OpDef{x_vcmp, C_VALU, (uint16_t)((T_I32 << 8) | e.p), "Uii"};
      t[std::string("v_cmp_") + e.n + "_u32" + suf] = OpDef{x_vcmp, C_VALU, (uint16_t)((T_U32 << 8) | e.p), "Uuu"};
      t[std::string("v_cmp_") + e.n + "_u16" + suf] = OpDef{x_vcmp, C_VALU, (uint16_t)((T_U16 << 8) | e.p), "Uuu"};
      t[std::string("v_cmp_") + e.n + "_i16" + suf] = OpDef{x_vcmp, C_VALU, (uint16_t)((T_I16 << 8) | e.p), "Uii"};
      t[std::string("v_cmp_") + e.n + "_u64" + suf] = OpDef{x_vcmp, C_VALU, (uint16_t)((T_U64 << 8) | e.p), "UUU"};
      t[std::string("v_cmp_") + e.n + "_i64" + suf] = OpDef{x_vcmp, C_VALU, (uint16_t)((T_I64 << 8) | e.p), "UUU"};
    }
  // ---- scalar
  add("s_mov_b32", x_salu, C_SALU, S_MOV32, "bb");
  add("s_mov_b64", x_salu, C_SALU, S_MOV64, "UU");
  add("s_movk_i32", x_salu, C_SALU, S_MOVK, "ii");
  add("s_add_i32", x_salu, C_SALU, S_ADD_I32, "iii");
  add("s_add_u32", x_salu, C_SALU, S_ADD_U32, "uuu");
  add("s_addc_u32", x_salu, C_SALU, S_ADDC_U32, "uuu");
  add("s_sub_u32", x_salu, C_SALU, S_SUB_U32, "uuu");
  add("s_subb_u32", x_salu, C_SALU, S_SUBB_U32, "uuu");
  add("s_addk_i32", x_salu, C_SALU, S_ADDK, "ii");
  add("s_sub_i32", x_salu, C_SALU, S_SUB_I32, "iii");
  add("s_mul_i32", x_salu, C_SALU, S_MUL_I32, "iii");
  add("s_mul_hi_u32", x_salu, C_SALU, S_MUL_HI_U32, "uuu");
  add("s_mul_hi_i32", x_salu, C_SALU, S_MUL_HI_I32, "iii");
  add("s_memrealtime s_memtime", x_salu, C_SMEM, S_MEMTIME, "U");
  add("s_and_b32", x_salu, C_SALU, S_AND32, "bbb");
  add("s_and_b64", x_salu, C_SALU, S_AND64, "UUU");
  add("s_or_b32", x_salu, C_SALU, S_OR32, "bbb");
  add("s_or_b64", x_salu, C_SALU, S_OR64, "UUU");
  add("s_xor_b32", x_salu, C_SALU, S_XOR32, "bbb");
  add("s_xor_b64", x_salu, C_SALU, S_XOR64, "UUU");
  add("s_xnor_b64", x_salu, C_SALU, S_XNOR64, "UUU");
  add("s_nand_b64", x_salu, C_SALU, S_NAND64, "UUU");
  add("s_nor_b64", x_salu, C_SALU, S_NOR64, "UUU");
  add("s_andn2_b64", x_salu, C_SALU, S_ANDN2_64, "UUU");
  add("s_andn2_b32", x_salu, C_SALU, S_ANDN2_32, "bbb");
  add("s_orn2_b64", x_salu, C_SALU, S_ORN2_64, "UUU");
  add("s_not_b64", x_salu, C_SALU, S_NOT64, "UU");
  add("s_not_b32", x_salu, C_SALU, S_NOT32, "bb");
  add("s_lshl_b32", x_salu, C_SALU, S_LSHL32, "bbu");
  add("s_lshl_b64", x_salu, C_SALU, S_LSHL64, "UUu");
  add("s_lshr_b32", x_salu, C_SALU, S_LSHR32, "bbu");
  add("s_lshr_b64", x_salu, C_SALU, S_LSHR64, "UUu");
  add("s_ashr_i32", x_salu, C_SALU, S_ASHR32, "iiu");
  add("s_bfm_b32", x_salu, C_SALU, S_BFM32, "buu");
  add("s_bfe_u32", x_salu, C_SALU, S_BFE_U32, "uuu");
  add("s_bfe_i32", x_salu, C_SALU, S_BFE_I32, "iiu");
  add("s_bfe_i64", x_salu, C_SALU, S_BFE_I64, "UUu");
  add("s_bfe_u64", x_salu, C_SALU, S_BFE_U64, "UUu");
  add("s_mulk_i32", x_salu, C_SALU, S_MULK, "ii");
  add("s_getpc_b64", x_salu, C_SALU, S_GETPC, "U");
  add("s_brev_b32", x_salu, C_SALU, S_BREV32, "bb");
  add("s_bcnt1_i32_b64", x_salu, C_SALU, S_BCNT1_64, "iU");
  add("s_bcnt1_i32_b32", x_salu, C_SALU, S_BCNT1_32, "ib");
  add("s_ff1_i32_b64", x_salu, C_SALU, S_FF1_64, "iU");
  add("s_ff1_i32_b32", x_salu, C_SALU, S_FF1_32, "ib");
  add("s_min_u32", x_salu, C_SALU, S_MIN_U32, "uuu");
  add("s_max_u32", x_salu, C_SALU, S_MAX_U32, "uuu");
  add("s_min_i32", x_salu, C_SALU, S_MIN_I32, "iii");
  add("s_max_i32", x_salu, C_SALU, S_MAX_I32, "iii");
  add("s_abs_i32", x_salu, C_SALU, S_ABS_I32, "ii");
  add("s_sext_i32_i16", x_salu, C_SALU, S_SEXT_I32_I16, "ii");
  add("s_cselect_b32", x_salu, C_SALU, S_CSELECT32, "bbb");
  add("s_cselect_b64", x_salu, C_SALU, S_CSELECT64, "UUU");
  add("s_and_saveexec_b64", x_salu, C_SALU, S_AND_SAVEEXEC, "UU");
  add("s_or_saveexec_b64", x_salu, C_SALU, S_OR_SAVEEXEC, "UU");
  add("s_andn2_saveexec_b64", x_salu, C_SALU, S_ANDN2_SAVEEXEC, "UU");
  add("s_getreg_b32", x_salu, C_SALU, S_GETREG, "bb");
  struct { const char *n; int p; } sp[] = {{"eq", SC_EQ}, {"lg", SC_LG}, {"gt", SC_GT}, {"ge", SC_GE}, {"lt", SC_LT}, {"le", SC_LE}};
  for (auto &e : sp) {
    t[std::string("s_cmp_") + e.n + "_u32"] = OpDef{x_scmp, C_SALU, (uint16_t)((T_U32 << 8) | e.p), "uu"};
    t[std::string("s_cmp_") + e.n + "_i32"] = OpDef{x_scmp, C_SALU, (uint16_t)((T_I32 << 8) | e.p), "ii"};
    t[std::string("s_cmp_") + e.n + "_u64"] = OpDef{x_scmp, C_SALU, (uint16_t)((T_U64 << 8) | e.p), "UU"};
    t[std::string("s_cmpk_") + e.n + "_u32"] = OpDef{x_scmp, C_SALU, (uint16_t)((T_U32 << 8) | e.p), "uu"};
    t[std::string("s_cmpk_") + e.n + "_i32"] = OpDef{x_scmp, C_SALU, (uint16_t)((T_I32 << 8) | e.p), "ii"};
  }
  add("s_bitcmp0_b32", x_scmp, C_SALU, (T_U32 << 8) | SC_BITCMP0, "bu");
  add("s_bitcmp1_b32", x_scmp, C_SALU, (T_U32 << 8) | SC_BITCMP1, "bu");
  add("s_bitcmp0_b64", x_scmp, C_SALU, (T_U64 << 8) | SC_BITCMP0, "Uu");
  add("s_bitcmp1_b64", x_scmp, C_SALU, (T_U64 << 8) | SC_BITCMP1, "Uu");
  add("v_trig_preop_f64", x_unmodelled, C_VALU, 0, "ddu");
  add("s_flbit_i32_b32", x_salu, C_SALU, S_FLBIT32, "ib");
  add("s_flbit_i32_b64", x_salu, C_SALU, S_FLBIT64, "iU");
  add("s_branch", x_branch, C_BRANCH, B_ALWAYS, "-");
  add("s_cbranch_scc0", x_branch, C_BRANCH, B_SCC0, "-");
  add("s_cbranch_scc1", x_branch, C_BRANCH, B_SCC1, "-");
  add("s_cbranch_vccz", x_branch, C_BRANCH, B_VCCZ, "-");
  add("s_cbranch_vccnz", x_branch, C_BRANCH, B_VCCNZ, "-");
  add("s_cbranch_execz", x_branch, C_BRANCH, B_EXECZ, "-");
  add("s_cbranch_execnz", x_branch, C_BRANCH, B_EXECNZ, "-");
  add("s_endpgm", x_endpgm, C_BRANCH, 0, "");
  add("s_waitcnt s_nop s_waitcnt_vscnt s_waitcnt_depctr s_setprio s_inst_prefetch s_clause s_delay_alu s_set_gpr_idx_off s_setreg_b32 s_setreg_imm32_b32 s_icache_inv s_dcache_wb buffer_wbl2 buffer_inv s_dcache_inv", x_nop, C_WAIT, 0, "--");
  add("s_sleep", x_sleep, C_WAIT, 0, "-");
  add("s_barrier", x_barrier, C_WAIT, 0, "");
  add("s_load_dword", x_sload, C_SMEM, 1, "bUu");
  add("s_load_dwordx2", x_sload, C_SMEM, 2, "bUu");
  add("s_load_dwordx4", x_sload, C_SMEM, 4, "bUu");
  add("s_load_dwordx8", x_sload, C_SMEM, 8, "bUu");
  add("s_load_dwordx16", x_sload, C_SMEM, 16, "bUu");
  // ---- LDS
  auto L = [](int kind, unsigned words) { return (uint16_t)(kind | (words << 8)); };
  add("ds_read_b32", x_lds, C_LDS, L(L_READ, 1), "bb");
  add("ds_read_b64", x_lds, C_LDS, L(L_READ, 2), "bb");
  add("ds_read_b96", x_lds, C_LDS, L(L_READ, 3), "bb");
  add("ds_read_b128", x_lds, C_LDS, L(L_READ, 4), "bb");
  add("ds_read_u8", x_lds, C_LDS, L(L_READ_U8, 1), "bb");
  add("ds_read_u16", x_lds, C_LDS, L(L_READ_U16, 1), "bb");
  add("ds_write_b32", x_lds, C_LDS, L(L_WRITE, 1), "bb");
  add("ds_write_b64", x_lds, C_LDS, L(L_WRITE, 2), "bb");
  add("ds_write_b96", x_lds, C_LDS, L(L_WRITE, 3), "bb");
  add("ds_write_b128", x_lds, C_LDS, L(L_WRITE, 4), "bb");
  add("ds_write_b8", x_lds, C_LDS, L(L_WRITE_B8, 1), "bb");
  add("ds_write_b16", x_lds, C_LDS, L(L_WRITE_B16, 1), "bb");
  add("ds_read2_b32", x_lds, C_LDS, L(L_READ2, 1), "bb");
  add("ds_read2_b64", x_lds, C_LDS, L(L_READ2, 2), "bb");
  add("ds_read2st64_b32", x_lds, C_LDS, L(L_READ2ST64, 1), "bb");
  add("ds_read2st64_b64", x_lds, C_LDS, L(L_READ2ST64, 2), "bb");
  add("ds_write2_b32", x_lds, C_LDS, L(L_WRITE2, 1), "bbb");
  add("ds_write2_b64", x_lds, C_LDS, L(L_WRITE2, 2), "bbb");
  add("ds_write2st64_b32", x_lds, C_LDS, L(L_WRITE2ST64, 1), "bbb");
  add("ds_write2st64_b64", x_lds, C_LDS, L(L_WRITE2ST64, 2), "bbb");
  add("ds_add_rtn_u32", x_lds, C_LDS, L(L_ADD_RTN, 1), "bbb");
  add("ds_add_rtn_u64", x_lds, C_LDS, L(L_ADD_RTN, 2), "bbU");
  add("ds_add_u32", x_lds, C_LDS, L(L_ADD, 1), "bb");
  add("ds_add_u64", x_lds, C_LDS, L(L_ADD, 2), "bU");
  add("ds_max_rtn_u32", x_lds, C_LDS, L(L_MAX_RTN_U32, 1), "bbb");
  add("ds_min_rtn_u32", x_lds, C_LDS, L(L_MIN_RTN_U32, 1), "bbb");
  add("ds_or_rtn_b32", x_lds, C_LDS, L(L_OR_RTN, 1), "bbb");
  add("ds_and_rtn_b32", x_lds, C_LDS, L(L_AND_RTN, 1), "bbb");
  add("ds_max_u32", x_lds, C_LDS, L(L_MAX_U32, 1), "bb");
  add("ds_min_i32", x_lds, C_LDS, L(L_MIN_I32, 1), "bb");
  add("ds_max_i32", x_lds, C_LDS, L(L_MAX_I32, 1), "bb");
  add("ds_min_u32", x_lds, C_LDS, L(L_MIN_U32, 1), "bb");
  add("ds_or_b32", x_lds, C_LDS, L(L_OR, 1), "bb");
  add("ds_and_b32", x_lds, C_LDS, L(L_AND, 1), "bb");
  add("ds_cmpst_rtn_b32", x_lds, C_LDS, L(L_CMPST_RTN, 1), "bbbb");
  add("ds_wrxchg_rtn_b32", x_lds, C_LDS, L(L_WRXCHG_RTN, 1), "bbb");
  add("ds_wrxchg_rtn_b64", x_lds, C_LDS, L(L_WRXCHG_RTN, 2), "bbU");
  add("ds_bpermute_b32", x_lds, C_LDS, L(L_BPERMUTE, 1), "bbb");
  // ---- flat / global / scratch
  auto V = [](int kind, unsigned words, bool global) { return (uint16_t)(kind | (words << 8) | (global ? 1u << 15 : 0u)); };
  for (int g = 0; g < 2; ++g) {
    const std::string pre = g ? "global_" : "flat_";
    const char *names[][2] = {{"load_dword", "1"}, {"load_dwordx2", "2"}, {"load_dwordx3", "3"}, {"load_dwordx4", "4"}};
    for (auto &n : names) t[pre + n[0]] = OpDef{x_vmem, C_VMEM, V(VM_LOAD, (unsigned)atoi(n[1]), g), "bUU"};
    const char *snames[][2] = {{"store_dword", "1"}, {"store_dwordx2", "2"}, {"store_dwordx3", "3"}, {"store_dwordx4", "4"}};
    for (auto &n : snames) t[pre + n[0]] = OpDef{x_vmem, C_VMEM, V(VM_STORE, (unsigned)atoi(n[1]), g), "UbU"};
    t[pre + "load_ubyte"] = OpDef{x_vmem, C_VMEM, V(VM_LOAD_U8, 1, g), "bUU"};
    t[pre + "load_ushort"] = OpDef{x_vmem, C_VMEM, V(VM_LOAD_U16, 1, g), "bUU"};
    t[pre + "load_sbyte"] = OpDef{x_vmem, C_VMEM, V(VM_LOAD_I8, 1, g), "bUU"};
    t[pre + "load_sshort"] = OpDef{x_vmem, C_VMEM, V(VM_LOAD_I16, 1, g), "bUU"};
    t[pre + "store_byte"] = OpDef{x_vmem, C_VMEM, V(VM_STORE_B8, 1, g), "UbU"};
    t[pre + "store_short"] = OpDef{x_vmem, C_VMEM, V(VM_STORE_B16, 1, g), "UbU"};
    t[pre + "atomic_add"] = OpDef{x_vmem, C_VMEM, V(VM_ATOMIC_ADD, 1, g), "bbbb"};
    t[pre + "atomic_add_x2"] = OpDef{x_vmem, C_VMEM, V(VM_ATOMIC_ADD, 2, g), "bbbb"};
    t[pre + "atomic_umax"] = OpDef{x_vmem, C_VMEM, V(VM_ATOMIC_UMAX, 1, g), "bbbb"};
    t[pre + "atomic_umax_x2"] = OpDef{x_vmem, C_VMEM, V(VM_ATOMIC_UMAX, 2, g), "bbbb"};
    t[pre + "atomic_umin"] = OpDef{x_vmem, C_VMEM, V(VM_ATOMIC_UMIN, 1, g), "bbbb"};
    t[pre + "atomic_or"] = OpDef{x_vmem, C_VMEM, V(VM_ATOMIC_OR, 1, g), "bbbb"};
    t[pre + "atomic_and"] = OpDef{x_vmem, C_VMEM, V(VM_ATOMIC_AND, 1, g), "bbbb"};
    t[pre + "atomic_swap"] = OpDef{x_vmem, C_VMEM, V(VM_ATOMIC_SWAP, 1, g), "bbbb"};
    t[pre + "atomic_swap_x2"] = OpDef{x_vmem, C_VMEM, V(VM_ATOMIC_SWAP, 2, g), "bbbb"};
    t[pre + "atomic_cmpswap"] = OpDef{x_vmem, C_VMEM, V(VM_ATOMIC_CMPSWAP, 1, g), "bbbb"};
  }
  for (int wds = 1; wds <= 4; ++wds) {
    const std::string suf = wds == 1 ? "dword" : ("dwordx" + std::to_string(wds));
    t["scratch_load_" + suf] = OpDef{x_scratch, C_VMEM, (uint16_t)(VM_LOAD | (wds << 8)), "bbb"};
    t["scratch_store_" + suf] = OpDef{x_scratch, C_VMEM, (uint16_t)(VM_STORE | (wds << 8)), "bbb"};
  }
  return t;
}

static std::string trim(const std::string &s) {
  size_t a = 0, b = s.size();
  while (a < b && isspace((unsigned char)s[a])) ++a;
  while (b > a && isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}
static bool parse_reg(const std::string &t, char pre, uint16_t &reg, uint8_t &n) {
  if (t.size() < 2 || t[0] != pre) return false;
  if (t[1] == '[') {
    unsigned a, b;
    if (sscanf(t.c_str() + 2, "%u:%u]", &a, &b) != 2) return false;
    reg = (uint16_t)a;
    n = (uint8_t)(b - a + 1);
    return true;
  }
  if (!isdigit((unsigned char)t[1])) return false;
  for (size_t i = 1; i < t.size(); ++i)
    if (!isdigit((unsigned char)t[i])) return false;
  reg = (uint16_t)atoi(t.c_str() + 1);
  n = 1;
  return true;
}
static const std::map<std::string, std::vector<uint8_t> *> *g_data_objs = nullptr;
static Operand parse_operand(std::string t, char ty, const std::map<std::string, int> &labels, bool &ok) {
  Operand o;
  ok = true;
  t = trim(t);
  if (!t.empty() && t[0] == '-' && t.size() > 1 && (t[1] == 'v' || t[1] == 's' || t[1] == '|')) {
    o.neg = true;
    t = t.substr(1);
  }
  if (t.size() > 2 && t[0] == '|' && t.back() == '|') {
    o.abs = true;
    t = t.substr(1, t.size() - 2);
  }
  if (parse_reg(t, 'v', o.reg, o.n)) { o.kind = K_VGPR; return o; }
  if (parse_reg(t, 's', o.reg, o.n)) { o.kind = K_SGPR; return o; }
  if (t == "vcc") { o.kind = K_VCC; o.n = 2; return o; }
  if (t == "vcc_lo") { o.kind = K_VCC_LO; return o; }
  if (t == "vcc_hi") { o.kind = K_VCC_HI; return o; }
  if (t == "exec") { o.kind = K_EXEC; o.n = 2; return o; }
  if (t == "exec_lo") { o.kind = K_EXEC_LO; return o; }
  if (t == "exec_hi") { o.kind = K_EXEC_HI; return o; }
  if (t == "scc") { o.kind = K_SCC; return o; }
  if (t == "m0") { o.kind = K_M0; return o; }
  if (t == "off") { o.kind = K_OFF; return o; }
  if (t == "null") { o.kind = K_NULL; return o; }
  if (t == "src_shared_base") { o.kind = K_SHARED_BASE; o.n = 2; return o; }
  if (t == "src_private_base") { o.kind = K_PRIVATE_BASE; o.n = 2; return o; }
  if (t.compare(0, 6, "hwreg(") == 0) { o.kind = K_HWREG; return o; }
  if (t.compare(0, 2, ".L") == 0) {
    auto it = labels.find(t);
    if (it == labels.end()) { ok = false; return o; }
    o.kind = K_LABEL;
    o.label = it->second;
    return o;
  }
  // a constant
  o.kind = K_IMM;
  {
    const size_t at = t.find("@rel32@");
    if (at != std::string::npos) { // SYM@rel32@lo+4 / SYM@rel32@hi+12 behind an s_getpc_b64 (which gives 0 here): the halves of SYM's address
      const std::string sym = t.substr(0, at);
      const bool hi = t.compare(at + 7, 2, "hi") == 0;
      if (!g_data_objs || !g_data_objs->count(sym)) { ok = false; return o; }
      const uint64_t a = (uint64_t)(uintptr_t)g_data_objs->at(sym)->data();
      o.imm = hi ? (a >> 32) : (a & 0xFFFFFFFFull);
      return o;
    }
  }
  const char *c = t.c_str();
  char *end = nullptr;
  if (t.compare(0, 2, "0x") == 0 || t.compare(0, 3, "-0x") == 0) {
    o.is_hex = true;
    const long long v = strtoll(c, &end, 16);
    if (*end) { ok = false; return o; }
    o.imm = (uint64_t)v;
  } else if (t.find_first_of(".eE") != std::string::npos && t.find_first_not_of("0123456789.eE+-") == std::string::npos) {
    o.is_float_tok = true;
    o.fval = strtod(c, &end);
    if (*end) { ok = false; return o; }
  } else {
    const long long v = strtoll(c, &end, 10);
    if (*end || t.empty()) { ok = false; return o; }
    o.imm = (uint64_t)v;
  }
  // typed value
  switch (ty) {
  case 'd':
    if (o.is_float_tok) o.imm = f64_bits(o.fval);
    else if (o.is_hex) o.imm = (o.imm & 0xFFFFFFFFull) << 32; // a 32-bit literal is the HIGH half of a 64-bit float operand
    break;                                                 // an inline integer stays the (sign-extended) 64-bit integer
  case 'f':
    if (o.is_float_tok) o.imm = f32_bits((float)o.fval);
    else o.imm &= 0xFFFFFFFFull;
    break;
  case 'U':
    if (o.is_float_tok) o.imm = f64_bits(o.fval);
    else if (o.is_hex) o.imm &= 0xFFFFFFFFull;
    break;
  default:
    if (o.is_float_tok) o.imm = f32_bits((float)o.fval);
    else o.imm &= 0xFFFFFFFFull;
    break;
  }
  return o;
}
// splits "a, b, c mod:1 mod2" into operand tokens (top-level commas) and trailing modifier words
static void split_operands(const std::string &rest, std::vector<std::string> &ops, std::vector<std::string> &mods) {
  std::string cur;
  int depth = 0;
  bool bar = false;
  std::vector<std::string> parts;
  for (char ch : rest) {
    if (ch == '[' || ch == '(') ++depth;
    if (ch == ']' || ch == ')') --depth;
    if (ch == '|') bar = !bar;
    if (ch == ',' && depth == 0 && !bar) {
      parts.push_back(trim(cur));
      cur.clear();
    } else cur += ch;
  }
  if (!trim(cur).empty()) parts.push_back(trim(cur));
  // the last part may carry modifiers after whitespace
  for (size_t i = 0; i < parts.size(); ++i) {
    if (i + 1 < parts.size()) { ops.push_back(parts[i]); continue; }
    // words of the last part (no split inside brackets / parentheses): the operand, then modifiers
    std::vector<std::string> words;
    std::string wd;
    int d2 = 0;
    for (char ch : parts[i]) {
      if (ch == '[' || ch == '(') ++d2;
      if (ch == ']' || ch == ')') --d2;
      if (isspace((unsigned char)ch) && d2 == 0) {
        if (!wd.empty()) words.push_back(wd);
        wd.clear();
      } else wd += ch;
    }
    if (!wd.empty()) words.push_back(wd);
    bool first = true;
    for (const std::string &tok : words) {
      const bool is_mod = (tok.find(':') != std::string::npos && tok[0] != 'v' && tok[0] != 's' && tok[0] != '-' && tok[0] != '|') || tok == "sc0" || tok == "sc1" || tok == "nt" ||
                          tok == "glc" || tok == "slc" || tok == "clamp";
      if (first && !is_mod) ops.push_back(tok);
      else mods.push_back(tok);
      first = false;
    }
  }
}

static std::mutex g_mu;
static std::vector<std::unique_ptr<Kernel>> g_kernels;
static std::map<std::string, Kernel *> g_by_name;
static bool g_loaded = false;

static void load_file(const std::string &path) {
  std::ifstream f(path);
  if (!f) {
    fprintf(stderr, "isa: cannot read %s\n", path.c_str());
    abort();
  }
  std::vector<std::string> lines;
  for (std::string l; std::getline(f, l);) lines.push_back(l);
  auto &ops = optable();
  std::map<uint32_t, std::string> files;
  // descriptors: .amdhsa_kernel NAME ... .end_amdhsa_kernel
  std::map<std::string, std::map<std::string, long>> desc;
  for (size_t i = 0; i < lines.size(); ++i) {
    const std::string t = trim(lines[i]);
    if (t.compare(0, 6, ".file\t") == 0 || t.compare(0, 6, ".file ") == 0) {
      unsigned no;
      char a[1024], b[1024];
      if (sscanf(t.c_str() + 6, "%u \"%1023[^\"]\" \"%1023[^\"]\"", &no, a, b) == 3) files[no] = b;
      else if (sscanf(t.c_str() + 6, "%u \"%1023[^\"]\"", &no, a) == 2) files[no] = a;
    }
    if (t.compare(0, 15, ".amdhsa_kernel ") == 0) {
      const std::string name = trim(t.substr(15));
      for (size_t j = i + 1; j < lines.size(); ++j) {
        const std::string u = trim(lines[j]);
        if (u == ".end_amdhsa_kernel") break;
        char key[128];
        long val;
        if (sscanf(u.c_str(), ".amdhsa_%127s %ld", key, &val) == 2) desc[name][key] = val;
      }
    }
  }
  // constant data objects (.rodata tables the code reaches through s_getpc + @rel32 relocations): NAME: .long / .quad / ... .size NAME
  static std::map<std::string, std::vector<uint8_t> *> data_objs; // (kept for the life of the process: the code holds their addresses)
  for (size_t i = 0; i < lines.size(); ++i) {
    const std::string t = trim(lines[i]);
    if (t.compare(0, 6, ".type\t") != 0 && t.compare(0, 6, ".type ") != 0) continue;
    if (t.find("@object") == std::string::npos) continue;
    const std::string name = trim(t.substr(6, t.find(',') - 6));
    size_t j = i + 1;
    while (j < lines.size() && trim(lines[j]) != name + ":") {
      if (trim(lines[j]).compare(0, 5, ".type") == 0) break;
      ++j;
    }
    if (j >= lines.size() || trim(lines[j]) != name + ":") continue;
    auto *buf = new std::vector<uint8_t>();
    for (size_t q = j + 1; q < lines.size(); ++q) {
      std::string u = lines[q];
      const size_t sc = u.find(';');
      if (sc != std::string::npos) u = u.substr(0, sc);
      u = trim(u);
      if (u.compare(0, 5, ".size") == 0) break;
      auto push = [&](uint64_t v, int n) { for (int b = 0; b < n; ++b) buf->push_back((uint8_t)(v >> (8 * b))); };
      if (u.compare(0, 5, ".long") == 0) push(strtoull(u.c_str() + 5, nullptr, 0), 4);
      else if (u.compare(0, 5, ".quad") == 0) push(strtoull(u.c_str() + 5, nullptr, 0), 8);
      else if (u.compare(0, 6, ".short") == 0) push(strtoull(u.c_str() + 6, nullptr, 0), 2);
      else if (u.compare(0, 5, ".byte") == 0) push(strtoull(u.c_str() + 5, nullptr, 0), 1);
      else if (u.compare(0, 5, ".zero") == 0) push(0, 0), buf->resize(buf->size() + strtoull(u.c_str() + 5, nullptr, 0), 0);
    }
    buf->resize(buf->size() + 64, 0);
    data_objs[name] = buf;
  }
  g_data_objs = &data_objs;
  // metadata: the implicit ("hidden") kernel arguments and where they sit
  std::map<std::string, std::vector<std::pair<uint32_t, std::string>>> hidden;
  {
    std::vector<std::pair<uint32_t, std::string>> cur;
    long off = -1;
    bool in_md = false;
    for (const std::string &l : lines) {
      const std::string t = trim(l);
      if (t == "amdhsa.kernels:") in_md = true;
      if (!in_md) continue;
      if (t.compare(0, 8, ".offset:") == 0 || t.compare(0, 10, "- .offset:") == 0) off = atol(t.c_str() + t.find(':') + 1);
      else if (t.compare(0, 12, ".value_kind:") == 0) {
        const std::string kind = trim(t.substr(12));
        if (kind.compare(0, 7, "hidden_") == 0 && off >= 0) cur.emplace_back((uint32_t)off, kind);
      } else if (t.compare(0, 6, ".name:") == 0) {
        hidden[trim(t.substr(6))] = cur;
        cur.clear();
      }
    }
  }
  for (size_t i = 0; i < lines.size(); ++i) {
    const std::string &l = lines[i];
    if (l.compare(0, 2, "_Z") != 0 || l.find(':') == std::string::npos) continue;
    const std::string name = l.substr(0, l.find(':'));
    if (!desc.count(name)) continue; // not a kernel
    size_t end = i + 1; // the function's last line: the one in front of its .Lfunc_end label (a kernel may hold several s_endpgm)
    while (end < lines.size() && lines[end].compare(0, 10, ".Lfunc_end") != 0) ++end;
    if (end >= lines.size()) continue;
    --end;
    auto k = std::make_unique<Kernel>();
    k->name = name;
    k->file = path;
    k->lds_static = (uint32_t)desc[name]["group_segment_fixed_size"];
    k->scratch_bytes = (uint32_t)desc[name]["private_segment_fixed_size"];
    k->kernarg_size = (uint32_t)desc[name]["kernarg_size"];
    k->src_files = files;
    k->hidden = hidden[name];
    if (desc[name]["user_sgpr_count"] != 2 || desc[name]["user_sgpr_kernarg_segment_ptr"] != 1 || desc[name]["system_sgpr_workgroup_id_x"] != 1 ||
        desc[name]["system_sgpr_workgroup_id_y"] != 0 || desc[name]["system_vgpr_workitem_id"] != 0) {
      fprintf(stderr, "isa: %s: register set-up other than {kernarg pointer, workgroup id x, work-item id x} is not modelled\n", name.c_str());
      continue;
    }
    // pass 1: labels -> instruction index
    std::map<std::string, int> labels;
    int idx = 0;
    for (size_t j = i + 1; j <= end; ++j) {
      std::string t = lines[j];
      const size_t sc = t.find(';');
      if (sc != std::string::npos) t = t.substr(0, sc);
      t = trim(t);
      if (t.empty() || t[0] == '.') {
        if (t.compare(0, 2, ".L") == 0 && t.back() == ':') labels[t.substr(0, t.size() - 1)] = idx;
        continue;
      }
      if (t.back() == ':') continue;
      ++idx;
    }
    // pass 2: instructions
    uint32_t cur_file = 0, cur_line = 0;
    bool bad = false;
    for (size_t j = i + 1; j <= end && !bad; ++j) {
      std::string t = lines[j];
      const size_t sc = t.find(';');
      if (sc != std::string::npos) t = t.substr(0, sc);
      t = trim(t);
      if (t.empty()) continue;
      if (t[0] == '.') {
        unsigned a, b;
        if (sscanf(t.c_str(), ".loc %u %u", &a, &b) == 2) { cur_file = a; cur_line = b; }
        continue;
      }
      if (t.back() == ':') continue;
      Inst in;
      in.line = (int)j + 1;
      in.text = t;
      in.src_file = cur_file;
      in.src_line = cur_line;
      const size_t sp = t.find_first_of(" \t");
      const std::string mn = sp == std::string::npos ? t : t.substr(0, sp);
      const std::string rest = sp == std::string::npos ? "" : trim(t.substr(sp));
      auto it = ops.find(mn);
      if (it == ops.end()) {
        fprintf(stderr, "isa: %s: opcode not implemented: %s  (%s line %d)\n", name.c_str(), t.c_str(), path.c_str(), in.line);
        bad = true;
        break;
      }
      in.fn = it->second.fn;
      in.cls = it->second.cls;
      in.sub = it->second.sub;
      std::vector<std::string> optok, mods;
      split_operands(rest, optok, mods);
      const char *sig = it->second.sig;
      const size_t siglen = strlen(sig);
      if (in.cls == C_WAIT && in.fn != x_barrier) { // s_waitcnt vmcnt(0) lgkmcnt(0), s_nop 0, s_sleep 4: nothing to evaluate
        optok.clear();
        mods.clear();
      }
      if (optok.size() > 5) { fprintf(stderr, "isa: too many operands: %s\n", t.c_str()); bad = true; break; }
      for (size_t q = 0; q < optok.size(); ++q) {
        bool ok;
        in.o[q] = parse_operand(optok[q], q < siglen ? sig[q] : 'b', labels, ok);
        if (!ok) { fprintf(stderr, "isa: cannot parse operand '%s' of: %s (line %d)\n", optok[q].c_str(), t.c_str(), in.line); bad = true; break; }
      }
      in.nops = (uint8_t)optok.size();
      for (const std::string &m : mods) {
        if (m.compare(0, 7, "offset:") == 0) in.offset = (int32_t)strtol(m.c_str() + 7, nullptr, 0);
        else if (m.compare(0, 8, "offset0:") == 0) in.offset0 = (int32_t)strtol(m.c_str() + 8, nullptr, 0);
        else if (m.compare(0, 8, "offset1:") == 0) in.offset1 = (int32_t)strtol(m.c_str() + 8, nullptr, 0);
        else if (m == "sc0" || m == "glc") in.sc0 = true;
        else if (m == "sc1" || m == "nt" || m == "slc") {}
        else if (m.compare(0, 7, "bitop3:") == 0) in.bitop3 = (uint32_t)strtoul(m.c_str() + 7, nullptr, 0);
        else if (m.compare(0, 11, "op_sel_hi:[") == 0) {
          in.op_sel_hi = 0;
          int k = 0;
          for (const char *c = m.c_str() + 11; *c && *c != ']'; ++c)
            if (*c == '0' || *c == '1') in.op_sel_hi |= (uint8_t)((*c - '0') << k++);
        } else if (m.compare(0, 9, "src0_sel:") == 0 || m.compare(0, 9, "src1_sel:") == 0 || m.compare(0, 8, "dst_sel:") == 0) {
          const std::string v = m.substr(m.find(':') + 1);
          uint8_t code = 6;
          if (v.compare(0, 5, "BYTE_") == 0) code = (uint8_t)(v[5] - '0');
          else if (v.compare(0, 5, "WORD_") == 0) code = (uint8_t)(4 + (v[5] - '0'));
          else if (v != "DWORD") { fprintf(stderr, "isa: %s not implemented: %s (line %d)\n", m.c_str(), t.c_str(), in.line); bad = true; }
          (m[0] == 'd' ? in.dsel : (m[3] == '0' ? in.sel0 : in.sel1)) = code;
        } else if (m.compare(0, 11, "dst_unused:") == 0) {
          in.dst_preserve = m.find("PRESERVE") != std::string::npos;
        } else if (m.compare(0, 8, "neg_lo:[") == 0 || m.compare(0, 8, "neg_hi:[") == 0) {
          uint8_t bits = 0;
          int k = 0;
          for (const char *c = m.c_str() + 8; *c && *c != ']'; ++c)
            if (*c == '0' || *c == '1') bits |= (uint8_t)((*c - '0') << k++);
          (m[5] == 'o' ? in.neg_lo : in.neg_hi) = bits;
        } else if (m.compare(0, 8, "op_sel:[") == 0) {
          int k = 0;
          for (const char *c = m.c_str() + 8; *c && *c != ']'; ++c)
            if (*c == '0' || *c == '1') in.op_sel |= (uint8_t)((*c - '0') << k++);
        } else {
          fprintf(stderr, "isa: modifier '%s' not implemented: %s (line %d)\n", m.c_str(), t.c_str(), in.line);
          bad = true;
        }
      }
      // shape fix-ups
      if (in.fn == x_vmem && (in.sub & 0xFF) >= VM_ATOMIC_ADD && (in.sub & 0xFF) != VM_LOAD_U8 && (in.sub & 0xFF) != VM_STORE_B8 && (in.sub & 0xFF) != VM_LOAD_U16 && (in.sub & 0xFF) != VM_STORE_B16) {
        // returning form iff the first operand is a VGPR destination in front of the address: decided by sc0 (the assembler requires it)
      }
      k->code.push_back(in);
    }
    if (bad) continue;
    k->hits.assign(k->code.size(), 0);
    g_by_name[name] = k.get();
    g_kernels.push_back(std::move(k));
  }
}

static void ensure_loaded() {
  if (g_loaded) return;
  g_loaded = true;
  const char *e = getenv("MGPU_EMU_ISA");
  if (!e || !*e) return;
  std::stringstream ss(e);
  std::string p;
  while (std::getline(ss, p, ':'))
    if (!p.empty()) load_file(p);
  if (getenv("MGPU_EMU_ISA_VERBOSE")) fprintf(stderr, "isa: %zu kernels loaded\n", g_kernels.size());
}

} // namespace isa

extern "C" {
unsigned long long isa_counters[16]; // [C_VALU .. C_OTHER] wave-instructions executed by interpreted launches, [8] launches, [9] waves
// per-instruction execution counts of every interpreted kernel: "<kernel>\t<.s line>\t<class>\t<count>\t<source file:line>\t<text>"
void isa_profile_dump(const char *path) {
  FILE *f = fopen(path, "w");
  if (!f) return;
  static const char *cn[] = {"valu", "salu", "branch", "lds", "vmem", "smem", "wait", "other"};
  for (auto &k : isa::g_kernels)
    for (size_t i = 0; i < k->code.size(); ++i)
      if (k->hits[i]) {
        const isa::Inst &in = k->code[i];
        auto it = k->src_files.find(in.src_file);
        fprintf(f, "%s\t%d\t%s\t%llu\t%s:%u\t%s\n", k->name.c_str(), in.line, cn[in.cls], (unsigned long long)k->hits[i], it == k->src_files.end() ? "?" : it->second.c_str(), in.src_line,
                in.text.c_str());
      }
  fclose(f);
}
void isa_profile_reset() {
  for (auto &k : isa::g_kernels) std::fill(k->hits.begin(), k->hits.end(), 0ull);
  memset(isa_counters, 0, sizeof(isa_counters));
}
}

namespace isa {

bool enabled() {
  std::lock_guard<std::mutex> lk(g_mu);
  ensure_loaded();
  return !g_kernels.empty();
}

// Runs the kernel named `mangled` if one of the loaded files has it: grid x block threads, `shmem` bytes of dynamic LDS behind the static part,
// the kernel arguments at `kernarg`.  Workgroups run one after the other (a persistent kernel's first workgroup takes all the work), the
// waves of a workgroup take turns: a wave runs until a barrier, an s_sleep, its end or kQuantum instructions.
bool run(const char *mangled, dim3 grid, dim3 block, size_t shmem, const void *kernarg, size_t kernarg_bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  ensure_loaded();
  auto it = g_by_name.find(mangled);
  if (it == g_by_name.end()) return false;
  Kernel &K = *it->second;
  const unsigned nthreads = block.x * block.y * block.z;
  if (block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1 || nthreads % 64) {
    fprintf(stderr, "isa: %s: only 1-D launches of whole waves are modelled\n", mangled);
    abort();
  }
  // kernarg: the explicit arguments followed by the implicit ones (zero: the kernels do not read them)
  std::vector<uint8_t> args((size_t)std::max<uint32_t>(K.kernarg_size, (uint32_t)kernarg_bytes) + 512, 0);
  memcpy(args.data(), kernarg, kernarg_bytes);
  for (const auto &h : K.hidden) { // the implicit arguments the code may read (grid-stride loops read the block counts)
    uint8_t *p = args.data() + h.first;
    auto put32 = [&](uint32_t v) { memcpy(p, &v, 4); };
    auto put16 = [&](uint16_t v) { memcpy(p, &v, 2); };
    if (h.second == "hidden_block_count_x") put32(grid.x);
    else if (h.second == "hidden_block_count_y") put32(grid.y);
    else if (h.second == "hidden_block_count_z") put32(grid.z);
    else if (h.second == "hidden_group_size_x") put16((uint16_t)block.x);
    else if (h.second == "hidden_group_size_y") put16((uint16_t)block.y);
    else if (h.second == "hidden_group_size_z") put16((uint16_t)block.z);
    else if (h.second == "hidden_grid_dims") put16(1);
  }
  const unsigned nw = nthreads / 64;
  // how long a wave runs before the next one of its workgroup gets its turn: 4096 instructions, or -- MGPU_EMU_ISA_QUANTUM=random[:seed] -- a
  // random 1 .. 512 per turn, which moves the waves against each other between every pair of barriers (a missing barrier or a racy LDS protocol
  // between waves then shows as a wrong frame sooner or later)
  static const char *qenv = getenv("MGPU_EMU_ISA_QUANTUM");
  static const bool qrandom = qenv && strncmp(qenv, "random", 6) == 0;
  static uint64_t qstate = qrandom && qenv[6] == ':' ? strtoull(qenv + 7, nullptr, 0) * 2654435761ull + 1 : 88172645463325252ull;
  size_t kQuantum = qenv && !qrandom && atol(qenv) > 0 ? (size_t)atol(qenv) : 4096;
  Machine M;
  M.k = &K;
  isa_counters[8] += 1;
  for (unsigned b = 0; b < grid.x; ++b) {
    M.block_id = b;
    M.lds.assign(kLdsBytes, 0xCD);
    if ((size_t)K.lds_static + shmem > 160 * 1024) { fprintf(stderr, "isa: %s asks for %zu bytes of LDS\n", mangled, (size_t)K.lds_static + shmem); abort(); }
    std::vector<Wave> waves(nw);
    for (unsigned wi = 0; wi < nw; ++wi) {
      Wave &w = waves[wi];
      w.id = (int)wi;
      memset(w.s, 0, sizeof(w.s));
      const uint64_t ka = (uint64_t)(uintptr_t)args.data();
      w.s[0] = (uint32_t)ka;
      w.s[1] = (uint32_t)(ka >> 32);
      w.s[2] = b;
      w.exec = ~0ull;
      w.v.assign((size_t)512 * 64, 0u);
      for (unsigned l = 0; l < 64; ++l) w.vr(0, l) = wi * 64 + l;
      w.scratch_stride = (K.scratch_bytes + 255u) & ~255u;
      w.scratch.assign((size_t)w.scratch_stride * 64, 0);
      isa_counters[9] += 1;
    }
    unsigned alive = nw;
    static const unsigned long long watchdog = getenv("MGPU_EMU_ISA_WATCHDOG") ? strtoull(getenv("MGPU_EMU_ISA_WATCHDOG"), nullptr, 0) : 0ull;
    unsigned long long executed = 0;
    while (alive) {
      if (watchdog && executed > watchdog) {
        fprintf(stderr, "isa: %s workgroup %u: %llu instructions and no end -- where the waves are:\n", mangled, b, executed);
        for (Wave &w : waves)
          fprintf(stderr, "  wave %d %s exec %016llx vcc %016llx scc %d  line %d: %s\n", w.id, w.done ? "done" : (w.at_barrier ? "at a barrier" : "running"), (unsigned long long)w.exec,
                  (unsigned long long)w.vcc, (int)w.scc, K.code[std::min(w.pc, K.code.size() - 1)].line, K.code[std::min(w.pc, K.code.size() - 1)].text.c_str());
        abort();
      }
      bool progressed = false;
      unsigned waiting = 0;
      for (Wave &w : waves) {
        if (w.done) continue;
        if (w.at_barrier) { ++waiting; continue; }
        progressed = true;
        M.yield = false;
        if (qrandom) {
          qstate ^= qstate << 13; qstate ^= qstate >> 7; qstate ^= qstate << 17;
          kQuantum = 1 + (size_t)(qstate % 512);
        }
        for (size_t n = 0; n < kQuantum && !M.yield; ++n) {
          const Inst &in = K.code[w.pc];
          K.hits[w.pc] += 1;
          M.counts[in.cls] += 1;
          in.fn(M, w, in);
          w.pc += 1;
          ++executed;
        }
        if (w.done) --alive;
      }
      if (alive && waiting == alive) { // every live wave has arrived: the barrier opens
        for (Wave &w : waves) w.at_barrier = false;
        progressed = true;
      }
      if (!progressed && alive) { fprintf(stderr, "isa: %s: deadlock (%u waves alive, %u at a barrier)\n", mangled, alive, waiting); abort(); }
    }
  }
  for (int c = 0; c < C_N; ++c) isa_counters[c] += M.counts[c];
  return true;
}

} // namespace isa
