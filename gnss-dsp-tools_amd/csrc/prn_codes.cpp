// PRN chip generators for the acquisition engine (host C++, no GPU needed).
//
// One table-driven generator per code family; every family is registered under the name of the
// reference module that defines it ("gps.ca" == gnsstools/gps/ca.py ...).  Chips are {0,1} bytes and
// must be BIT-EXACT with the reference; tests/test_native_cpu.py and
// tests/test_driver_visible_parity.py pin all 2355 PRNs against SHA-256 goldens generated from the reference
// (tests/golden/chips_sha256.json) and against the ICD known-answer vectors (tests/golden/icd_kat.json).
//
// Register convention used by every shift-register family below: bit i of `s` is stage x[i];
// one shift inserts the feedback at stage 0 and moves stage i-1 -> i (the reference's
// `[fb] + x[0:len-1]`, e.g. gnsstools/gps/ca.py:55-59).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gacq.h"

namespace {

struct PrnRow { int prn; long a, b, c; };
#include "icd_tables.inc"

struct ShiftReg {
  uint32_t s, mask, taps;
  ShiftReg(int len, uint32_t init, std::initializer_list<int> tap_list) : s(init), mask((1u << len) - 1), taps(0) {
    for (int t : tap_list) taps |= 1u << t;
  }
  int stage(int i) const { return (s >> i) & 1u; }
  void shift() { s = ((s << 1) | (uint32_t)__builtin_parity(s & taps)) & mask; }
};

const PrnRow* find_row(const PrnRow* t, int n, int prn) {
  for (int i = 0; i < n; i++) if (t[i].prn == prn) return &t[i];
  return nullptr;
}

typedef std::vector<uint8_t> Chips;

// ---- GPS C/A: G1 xor G2 delayed (gnsstools/gps/ca.py:55-104) -------------------------------
bool gen_gps_ca(int prn, Chips& out) {
  const PrnRow* r = find_row(T_gps_ca, N_gps_ca, prn);
  if (!r) return false;
  const int L = 1023;
  ShiftReg g1(10, 0x3ff, {9, 2}), g2(10, 0x3ff, {9, 8, 7, 5, 2, 1});
  std::vector<uint8_t> a(L), b(L);
  for (int i = 0; i < L; i++) { a[i] = g1.stage(9); b[i] = g2.stage(9); g1.shift(); g2.shift(); }
  out.resize(L);
  const int d = (int)r->a;
  for (int i = 0; i < L; i++) out[i] = a[i] ^ b[(i + L - d) % L];   // circular_shift(g2, d)
  return true;
}

// ---- GPS L5I / L5Q: XA (short-cycled) xor XB advanced (gnsstools/gps/l5i.py:73-107) --------
bool gen_gps_l5(const PrnRow* tab, int ntab, int prn, Chips& out) {
  const PrnRow* r = find_row(tab, ntab, prn);
  if (!r) return false;
  const int L = 10230;
  ShiftReg xa(13, 0x1fff, {12, 11, 9, 8});
  ShiftReg xb(13, 0x1fff, {12, 11, 7, 6, 5, 3, 2, 0});       // L5I and L5Q share XA and XB; only the XB advance (table column a) differs
  std::vector<uint8_t> b(8191);
  for (int i = 0; i < 8191; i++) { b[i] = xb.stage(12); xb.shift(); }
  out.resize(L);
  const uint32_t short_cycle = 0x1fff & ~(1u << 11);   // stages 0..10 and 12 set, stage 11 clear
  for (int i = 0; i < L; i++) {
    out[i] = xa.stage(12) ^ b[((int)r->a + i) % 8191];
    if (xa.s == short_cycle) xa.s = 0x1fff; else xa.shift();
  }
  return true;
}

// ---- GPS L2CM: 27-bit Galois register (gnsstools/gps/l2cm.py:40-50) --------------------------
bool gen_gps_l2cm(int prn, Chips& out) {
  const PrnRow* r = find_row(T_gps_l2cm, N_gps_l2cm, prn);
  if (!r) return false;
  uint32_t x = (uint32_t)r->a;
  out.resize(10230);
  for (int i = 0; i < 10230; i++) { out[i] = x & 1u; x = (x >> 1) ^ ((x & 1u) ? 0445112474u : 0u); }
  return true;
}

// ---- GPS L2CL: same 27-bit register as L2CM, 767250 chips (gnsstools/gps/l2cl.py:40-50) ----------
bool gen_gps_l2cl(int prn, Chips& out) {
  const PrnRow* r = find_row(T_gps_l2cl, N_gps_l2cl, prn);
  if (!r) return false;
  uint32_t x = (uint32_t)r->a;
  out.resize(767250);
  for (int i = 0; i < 767250; i++) { out[i] = x & 1u; x = (x >> 1) ^ ((x & 1u) ? 0445112474u : 0u); }
  return true;
}

// ---- Weil codes: GPS L1Cd/p (N=10223 + 7-chip insertion), BDS B1Cd/p (N=10243 truncated) -----
// Legendre sequence: L[i]=1 iff i is a non-zero quadratic residue mod N (gnsstools/gps/l1cd.py:59-62)
const std::vector<uint8_t>& legendre(int N) {
  static std::map<int, std::vector<uint8_t>> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(N);
  if (it != cache.end()) return it->second;
  std::vector<uint8_t> v(N, 0);
  for (long k = 1; k < N; k++) v[(k * k) % N] = 1;
  return cache.emplace(N, std::move(v)).first->second;
}

bool gen_weil_gps(const PrnRow* tab, int ntab, int prn, Chips& out) {
  const PrnRow* r = find_row(tab, ntab, prn);
  if (!r) return false;
  const int N = 10223;
  const auto& Lg = legendre(N);
  const int w = (int)r->a, p = (int)r->b;
  static const uint8_t expansion[7] = {0, 1, 1, 0, 1, 0, 0};
  out.clear();
  out.reserve(10230);
  for (int k = 0; k < p - 1; k++) out.push_back(Lg[k] ^ Lg[(k + w) % N]);
  for (int k = 0; k < 7; k++) out.push_back(expansion[k]);
  for (int k = p - 1; k < N; k++) out.push_back(Lg[k] ^ Lg[(k + w) % N]);
  return true;
}

bool gen_weil_bds(const PrnRow* tab, int ntab, int prn, Chips& out) {
  const PrnRow* r = find_row(tab, ntab, prn);
  if (!r) return false;
  const int N = 10243;
  const auto& Lg = legendre(N);
  const int w = (int)r->a, p = (int)r->b;
  out.resize(10230);
  for (int n = 0; n < 10230; n++) { const int k = (n + p - 1) % N; out[n] = Lg[k] ^ Lg[(k + w) % N]; }
  return true;
}

// ---- Galileo E5: two 14-stage registers (gnsstools/galileo/e5ai.py:48-88) ---------------------
bool gen_gal_e5(const PrnRow* tab, int ntab, std::initializer_list<int> t1, std::initializer_list<int> t2,
                int prn, Chips& out) {
  const PrnRow* r = find_row(tab, ntab, prn);
  if (!r) return false;
  ShiftReg r1(14, 0x3fff, t1), r2(14, (uint32_t)r->a & 0x3fff, t2);
  out.resize(10230);
  for (int i = 0; i < 10230; i++) { out[i] = r1.stage(13) ^ r2.stage(13); r1.shift(); r2.shift(); }
  return true;
}

// ---- BeiDou B1I (= B2I): two 11-stage registers, phase selector (gnsstools/beidou/b1i.py:27-56)
bool gen_bds_b1i(int prn, Chips& out) {
  const PrnRow* r = find_row(T_bds_b1i, N_bds_b1i, prn);
  if (!r) return false;
  const uint32_t init = 0x2aa;   // stages 0..10 = 0,1,0,1,0,1,0,1,0,1,0
  ShiftReg g1(11, init, {0, 6, 7, 8, 9, 10}), g2(11, init, {0, 1, 2, 3, 4, 7, 8, 10});
  out.resize(2046);
  for (int i = 0; i < 2046; i++) {
    int v = g1.stage(10) ^ g2.stage((int)r->a - 1) ^ g2.stage((int)r->b - 1);
    if (r->c) v ^= g2.stage((int)r->c - 1);
    out[i] = (uint8_t)v;
    g1.shift(); g2.shift();
  }
  return true;
}

// ---- BeiDou 13-stage pairs: B2ad/B2ap/B2bd/B2bp reset g1 after chip 8189; B3I short-cycles g1
bool gen_bds_13(const PrnRow* tab, int ntab, std::initializer_list<int> t1, std::initializer_list<int> t2,
                bool b3i, int prn, Chips& out) {
  const PrnRow* r = find_row(tab, ntab, prn);
  if (!r) return false;
  ShiftReg g1(13, 0x1fff, t1), g2(13, (uint32_t)r->a & 0x1fff, t2);
  const uint32_t b3i_cycle = 0x1fff & ~((1u << 11) | (1u << 12));   // stages 0..10 set, 11,12 clear
  out.resize(10230);
  for (int i = 0; i < 10230; i++) {
    out[i] = g1.stage(12) ^ g2.stage(12);
    if (b3i) { if (g1.s == b3i_cycle) g1.s = 0x1fff; else g1.shift(); }
    else     { if (i == 8189) g1.s = 0x1fff; else g1.shift(); }
    g2.shift();
  }
  return true;
}

// ---- GLONASS C/A: 9-stage m-sequence, output stage 6 (gnsstools/glonass/ca.py:10-21) ----------
bool gen_glo_ca(int /*prn*/, Chips& out) {
  ShiftReg x(9, 0x1ff, {8, 4});
  out.resize(511);
  for (int i = 0; i < 511; i++) { out[i] = x.stage(6); x.shift(); }
  return true;
}

// ---- GLONASS P: 25-stage register, taps 24 and 2, output stage 9, 5.11 M chips (gnsstools/glonass/p.py:10-21)
bool gen_glo_p(int /*prn*/, Chips& out) {
  ShiftReg x(25, 0x1ffffff, {24, 2});
  out.resize(5110000);
  for (int i = 0; i < 5110000; i++) { out[i] = x.stage(9); x.shift(); }
  return true;
}

// ---- GLONASS L3OC d/p: 14-stage g2 xor 7-stage register seeded MSB-first with n (+64 for pilot)
bool gen_glo_l3oc(bool pilot, int prn, Chips& out) {
  if (prn < 0 || prn > 63) return false;
  const int seed = pilot ? prn + 64 : prn;
  uint32_t s7 = 0;
  for (int i = 0; i < 7; i++) s7 |= (uint32_t)((seed >> (6 - i)) & 1) << i;
  const int g2_init_bits[14] = {0, 0, 1, 1, 0, 1, 0, 0, 1, 1, 1, 0, 0, 0};
  uint32_t s14 = 0;
  for (int i = 0; i < 14; i++) s14 |= (uint32_t)g2_init_bits[i] << i;
  ShiftReg g(7, s7, {6, 5}), g2(14, s14, {13, 12, 7, 3});
  out.resize(10230);
  for (int i = 0; i < 10230; i++) { out[i] = g.stage(6) ^ g2.stage(13); g.shift(); g2.shift(); }
  return true;
}

// ---- memory codes: packed chips in data/memcodes.bin (see tools/gen_icd_tables.py) -------------
extern "C" const unsigned char gacq_memcodes_blob[];
extern "C" const unsigned char gacq_memcodes_blob_end[];
asm(".section .rodata\n"
    ".balign 16\n"
    ".global gacq_memcodes_blob\n"
    "gacq_memcodes_blob:\n"
    ".incbin \"" GACQ_MEMCODES_PATH "\"\n"
    ".global gacq_memcodes_blob_end\n"
    "gacq_memcodes_blob_end:\n"
    ".byte 0\n"
    ".text\n");

struct MemFam { int nprn, L; const int32_t* ids; const unsigned char* bits; };

bool mem_family(const char* cname, MemFam& f) {
  const unsigned char* b = gacq_memcodes_blob;
  if (memcmp(b, "GMC1", 4) != 0) return false;
  uint32_t n; memcpy(&n, b + 4, 4);
  for (uint32_t i = 0; i < n; i++) {
    const unsigned char* e = b + 8 + i * 28;
    if (strncmp((const char*)e, cname, 12) == 0) {
      uint32_t v[4]; memcpy(v, e + 12, 16);
      f.nprn = (int)v[0]; f.L = (int)v[1];
      f.ids = (const int32_t*)(b + v[2]); f.bits = b + v[3];
      return true;
    }
  }
  return false;
}

bool gen_mem(const char* cname, int prn, Chips& out) {
  MemFam f;
  if (!mem_family(cname, f)) return false;
  const int nbytes = (f.L + 7) / 8;
  for (int k = 0; k < f.nprn; k++) if (f.ids[k] == prn) {
    const unsigned char* p = f.bits + (size_t)k * nbytes;
    out.resize(f.L);
    for (int i = 0; i < f.L; i++) out[i] = (p[i >> 3] >> (7 - (i & 7))) & 1;
    return true;
  }
  return false;
}

// ---- registry -----------------------------------------------------------------------------------
struct Family {
  const char* name; int code_length; double chip_rate;
  const PrnRow* tab; int ntab; const char* mem; int lo, hi;   // PRN domain: table, memory family or [lo,hi]
};

const Family FAMILIES[] = {
  {"gps.ca", 1023, 1023000, T_gps_ca, N_gps_ca, nullptr, 0, 0},
  {"gps.l5i", 10230, 10230000, T_gps_l5i, N_gps_l5i, nullptr, 0, 0},
  {"gps.l5q", 10230, 10230000, T_gps_l5q, N_gps_l5q, nullptr, 0, 0},
  {"gps.l2cm", 10230, 511500, T_gps_l2cm, N_gps_l2cm, nullptr, 0, 0},
  {"gps.l2cl", 767250, 511500, T_gps_l2cl, N_gps_l2cl, nullptr, 0, 0},
  {"gps.l1cd", 10230, 1023000, T_gps_l1cd, N_gps_l1cd, nullptr, 0, 0},
  {"gps.l1cp", 10230, 1023000, T_gps_l1cp, N_gps_l1cp, nullptr, 0, 0},
  {"galileo.e1b", 4092, 1023000, nullptr, 0, "gal_e1b", 0, 0},
  {"galileo.e1c", 4092, 1023000, nullptr, 0, "gal_e1c", 0, 0},
  {"galileo.e5ai", 10230, 10230000, T_gal_e5ai, N_gal_e5ai, nullptr, 0, 0},
  {"galileo.e5aq", 10230, 10230000, T_gal_e5aq, N_gal_e5aq, nullptr, 0, 0},
  {"galileo.e5bi", 10230, 10230000, T_gal_e5bi, N_gal_e5bi, nullptr, 0, 0},
  {"galileo.e5bq", 10230, 10230000, T_gal_e5bq, N_gal_e5bq, nullptr, 0, 0},
  {"galileo.e6b", 5115, 5115000, nullptr, 0, "gal_e6b", 0, 0},
  {"galileo.e6c", 5115, 5115000, nullptr, 0, "gal_e6c", 0, 0},
  {"beidou.b1i", 2046, 2046000, T_bds_b1i, N_bds_b1i, nullptr, 0, 0},
  {"beidou.b1cd", 10230, 1023000, T_bds_b1cd, N_bds_b1cd, nullptr, 0, 0},
  {"beidou.b1cp", 10230, 1023000, T_bds_b1cp, N_bds_b1cp, nullptr, 0, 0},
  {"beidou.b2ad", 10230, 10230000, T_bds_b2ad, N_bds_b2ad, nullptr, 0, 0},
  {"beidou.b2ap", 10230, 10230000, T_bds_b2ap, N_bds_b2ap, nullptr, 0, 0},
  {"beidou.b2bd", 10230, 10230000, T_bds_b2bd, N_bds_b2bd, nullptr, 0, 0},
  {"beidou.b2bp", 10230, 10230000, T_bds_b2bp, N_bds_b2bp, nullptr, 0, 0},
  {"beidou.b2bi", 10230, 10230000, nullptr, 0, "bds_b2bi", 0, 0},
  {"beidou.b2bq", 10230, 10230000, nullptr, 0, "bds_b2bq", 0, 0},
  {"beidou.b3i", 10230, 10230000, T_bds_b3i, N_bds_b3i, nullptr, 0, 0},
  {"glonass.ca", 511, 511000, nullptr, 0, nullptr, 0, 0},       // single code; PRN argument ignored (use 0)
  {"glonass.p", 5110000, 5110000, nullptr, 0, nullptr, 0, 0},      // single code; PRN argument ignored (use 0)
  {"glonass.l3ocd", 10230, 10230000, nullptr, 0, nullptr, 0, 63},
  {"glonass.l3ocp", 10230, 10230000, nullptr, 0, nullptr, 0, 63},
  {"xona.x1p", 1023, 1023000, nullptr, 0, "xona_x1p", 0, 0},
  {"xona.x1d", 1023, 1023000, nullptr, 0, "xona_x1d", 0, 0},
  {"xona.x5p", 10230, 10230000, nullptr, 0, "xona_x5p", 0, 0},
};
const int NFAM = (int)(sizeof(FAMILIES) / sizeof(FAMILIES[0]));

const Family* find_family(const char* name) {
  if (!name) return nullptr;
  for (int i = 0; i < NFAM; i++) if (strcmp(FAMILIES[i].name, name) == 0) return &FAMILIES[i];
  return nullptr;
}

bool generate(const Family& f, int prn, Chips& out) {
  const std::string n = f.name;
  if (n == "gps.ca") return gen_gps_ca(prn, out);
  if (n == "gps.l5i") return gen_gps_l5(T_gps_l5i, N_gps_l5i, prn, out);
  if (n == "gps.l5q") return gen_gps_l5(T_gps_l5q, N_gps_l5q, prn, out);
  if (n == "gps.l2cm") return gen_gps_l2cm(prn, out);
  if (n == "gps.l2cl") return gen_gps_l2cl(prn, out);
  if (n == "glonass.p") return gen_glo_p(prn, out);
  if (n == "gps.l1cd") return gen_weil_gps(T_gps_l1cd, N_gps_l1cd, prn, out);
  if (n == "gps.l1cp") return gen_weil_gps(T_gps_l1cp, N_gps_l1cp, prn, out);
  if (n == "beidou.b1cd") return gen_weil_bds(T_bds_b1cd, N_bds_b1cd, prn, out);
  if (n == "beidou.b1cp") return gen_weil_bds(T_bds_b1cp, N_bds_b1cp, prn, out);
  if (n == "galileo.e5ai") return gen_gal_e5(T_gal_e5ai, N_gal_e5ai, {13, 7, 5, 0}, {13, 11, 7, 6, 4, 3}, prn, out);
  if (n == "galileo.e5aq") return gen_gal_e5(T_gal_e5aq, N_gal_e5aq, {13, 7, 5, 0}, {13, 11, 7, 6, 4, 3}, prn, out);
  if (n == "galileo.e5bi") return gen_gal_e5(T_gal_e5bi, N_gal_e5bi, {13, 12, 10, 3}, {13, 11, 8, 7, 4, 1}, prn, out);
  if (n == "galileo.e5bq") return gen_gal_e5(T_gal_e5bq, N_gal_e5bq, {13, 12, 10, 3}, {13, 9, 8, 5, 4, 0}, prn, out);
  if (n == "beidou.b1i") return gen_bds_b1i(prn, out);
  if (n == "beidou.b2ad") return gen_bds_13(T_bds_b2ad, N_bds_b2ad, {0, 4, 10, 12}, {2, 4, 8, 10, 11, 12}, false, prn, out);
  if (n == "beidou.b2ap") return gen_bds_13(T_bds_b2ap, N_bds_b2ap, {2, 5, 6, 12}, {0, 4, 6, 7, 11, 12}, false, prn, out);
  if (n == "beidou.b2bd") return gen_bds_13(T_bds_b2bd, N_bds_b2bd, {0, 8, 9, 12}, {2, 3, 5, 8, 11, 12}, false, prn, out);
  if (n == "beidou.b2bp") return gen_bds_13(T_bds_b2bp, N_bds_b2bp, {0, 10, 11, 12}, {1, 7, 8, 9, 10, 12}, false, prn, out);
  if (n == "beidou.b3i") return gen_bds_13(T_bds_b3i, N_bds_b3i, {0, 2, 3, 12}, {0, 4, 5, 6, 8, 9, 11, 12}, true, prn, out);
  if (n == "glonass.ca") return gen_glo_ca(prn, out);
  if (n == "glonass.l3ocd") return gen_glo_l3oc(false, prn, out);
  if (n == "glonass.l3ocp") return gen_glo_l3oc(true, prn, out);
  if (f.mem) return gen_mem(f.mem, prn, out);
  return false;
}

// per-(family, prn) cache, like the reference's module-level `codes` dicts (gnsstools/gps/ca.py:99-104)
std::mutex g_cache_mu;
std::map<std::pair<const Family*, int>, Chips> g_cache;

const Chips* cached_chips(const Family& f, int prn) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  auto key = std::make_pair(&f, prn);
  auto it = g_cache.find(key);
  if (it != g_cache.end()) return &it->second;
  Chips c;
  if (!generate(f, prn, c) || (int)c.size() != f.code_length) return nullptr;
  return &g_cache.emplace(key, std::move(c)).first->second;
}

}  // namespace

extern "C" {

int gacq_code_count(void) { return NFAM; }

const char* gacq_code_name(int index) { return (index >= 0 && index < NFAM) ? FAMILIES[index].name : nullptr; }

int gacq_code_length(const char* code) {
  const Family* f = find_family(code);
  return f ? f->code_length : GACQ_ERR_UNKNOWN_CODE;
}

double gacq_code_chip_rate(const char* code) {
  const Family* f = find_family(code);
  return f ? f->chip_rate : -1.0;
}

int gacq_code_prns(const char* code, int* out, int cap) {
  const Family* f = find_family(code);
  if (!f) return GACQ_ERR_UNKNOWN_CODE;
  std::vector<int> ids;
  if (f->tab) for (int i = 0; i < f->ntab; i++) ids.push_back(f->tab[i].prn);
  else if (f->mem) { MemFam m; if (!mem_family(f->mem, m)) return GACQ_ERR_INTERNAL; for (int i = 0; i < m.nprn; i++) ids.push_back(m.ids[i]); }
  else for (int p = f->lo; p <= f->hi; p++) ids.push_back(p);
  for (int i = 0; i < (int)ids.size() && i < cap; i++) out[i] = ids[i];
  return (int)ids.size();
}

int gacq_code_chips(const char* code, int prn, uint8_t* out, int cap) {
  const Family* f = find_family(code);
  if (!f) return GACQ_ERR_UNKNOWN_CODE;
  const Chips* c = cached_chips(*f, prn);
  if (!c) return GACQ_ERR_BAD_PRN;
  if (cap < f->code_length) return GACQ_ERR_BAD_ARG;
  memcpy(out, c->data(), f->code_length);
  return f->code_length;
}

// Replica sampler: <sig>.code(prn,0,0,L/n,n) [* nco.boc11(0,0,L/n,n)]  (gnsstools/gps/ca.py:106-112,
// gnsstools/nco.py:12-19, call sites acquire-gps-l1.py:22-23, acquire-galileo-e1b.py:23-25).
// Index arithmetic is fp64 exactly as numpy does it: floor(incr*i) then mod L.
int gacq_code_replica(const char* code, int prn, int n, int boc, float* out) {
  const Family* f = find_family(code);
  if (!f) return GACQ_ERR_UNKNOWN_CODE;
  if (n <= 0 || !out) return GACQ_ERR_BAD_ARG;
  const Chips* c = cached_chips(*f, prn);
  if (!c) return GACQ_ERR_BAD_PRN;
  const int L = f->code_length;
  const double incr = (double)L / (double)n;
  for (int i = 0; i < n; i++) {
    const double pos = 0.0 + 0.0 + incr * (double)i;
    long idx = (long)std::floor(pos) % L;
    float v = 1.0f - 2.0f * (float)(*c)[idx];
    if (boc) {
      long b = (long)std::floor(pos * 2.0) % 2;
      v *= b ? 1.0f : -1.0f;
    }
    out[i] = v;
  }
  return n;
}

}  // extern "C"
