// Chunk files: `<chunk>/test_data.pkl` as the reference writes it -- a pickled dict of four or five lists
// of numpy arrays (MakeDataForOptimization/process_test_data.py:149-157), the heat-maps being what scipy.io.loadmat returned
// (:65-67): [H,W,J] arrays in FORTRAN order, float32 or float64 -- to the layout the optimiser reads (heat [F,H,W,J] f32 in HBM,
// optimizer.py:251-252), without building a Python object per array:
//
//   gem_pickle_scan        host, no GIL: a bounds-checked interpreter of the pickle opcodes such a file consists of (protocols 2-5)
//                          that RECORDS where every array's bytes lie in the file instead of copying them; it constructs nothing
//                          and calls nothing the file names (unlike pickle.load, a hostile file can make it do no more than
//                          return "unsupported")
//   gem_pickle_gather_f64  host: the small per-frame arrays (skeletons, camera poses) of one key as one dense float64 array
//   gem_heat_gather        device: picks the heat-maps out of an image of the file in HBM, undoing the Fortran order and the
//                          float64 of loadmat's arrays on the way (what `np.asarray(data['heatmap_list'])` + `.float()` +
//                          `.permute` amount to at optimizer.py:324,248,251), whatever the byte alignment of the payloads
#include "gem_internal.h"

#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace gem {
namespace {

// ------------------------------------------------------------------------------------------------------------ the opcode interpreter
enum Kind : uint8_t { K_NONE, K_BOOL, K_INT, K_FLOAT, K_STR, K_BYTES, K_GLOBAL, K_TUPLE, K_LIST, K_DICT, K_OBJ };

struct Node {
    Kind kind = K_NONE;
    int64_t i = 0;                 // K_BOOL / K_INT: value; K_STR / K_BYTES: offset of the data in the image
    int64_t n = 0;                 // K_STR / K_BYTES: length
    int32_t a = -1, b = -1;        // K_GLOBAL: module, name (K_STR nodes); K_OBJ: callable, arguments
    int32_t state = -1;            // K_OBJ: what BUILD handed over
    std::vector<int32_t> items;    // K_TUPLE / K_LIST: elements; K_DICT: key, value, key, value ...
};

constexpr size_t MAX_NODES = 1u << 23, MAX_MEMO = 1u << 23;          // (a chunk of 100 frames makes ~5 k objects)

struct Scanner {
    const uint8_t* p;
    int64_t len, pos = 0;
    std::vector<Node> nodes;
    std::vector<int32_t> stack, memo;
    std::vector<size_t> marks;
    size_t memo_len = 0;
    std::string why;

    bool fail(const std::string& m) { if (why.empty()) why = m + " at byte " + std::to_string(pos); return false; }
    bool need(int64_t k) { return (k >= 0 && pos + k <= len) ? true : fail("truncated pickle"); }
    uint64_t le(int k) { uint64_t v = 0; for (int i = 0; i < k; ++i) v |= (uint64_t)p[pos + i] << (8 * i); pos += k; return v; }
    int32_t make(Kind k) {
        if (nodes.size() >= MAX_NODES) { fail("too many objects"); return -1; }
        nodes.emplace_back();
        nodes.back().kind = k;
        return (int32_t)nodes.size() - 1;
    }
    bool push(int32_t id) { if (id < 0) return false; stack.push_back(id); return true; }
    bool pop(int32_t& id) {
        if (stack.empty() || (!marks.empty() && stack.size() <= marks.back())) return fail("stack underflow");
        id = stack.back(); stack.pop_back(); return true;
    }
    bool top(int32_t& id) {
        if (stack.empty() || (!marks.empty() && stack.size() <= marks.back())) return fail("stack underflow");
        id = stack.back(); return true;
    }
    bool pop_mark(std::vector<int32_t>& out) {
        if (marks.empty()) return fail("no MARK");
        const size_t m = marks.back();
        marks.pop_back();
        out.assign(stack.begin() + m, stack.end());
        stack.resize(m);
        return true;
    }
    bool counted(Kind k, int width) {                 // a length-prefixed run of bytes: recorded, not copied
        if (!need(width)) return false;
        const uint64_t n = le(width);
        if (n > (uint64_t)len || !need((int64_t)n)) return false;
        const int32_t id = make(k);
        if (id < 0) return false;
        nodes[id].i = pos; nodes[id].n = (int64_t)n;
        pos += (int64_t)n;
        return push(id);
    }
    bool integer(int64_t v) { const int32_t id = make(K_INT); if (id < 0) return false; nodes[id].i = v; return push(id); }
    bool put(size_t idx) {
        int32_t t;
        if (!top(t)) return false;
        if (idx >= MAX_MEMO) return fail("memo index out of range");
        if (idx >= memo.size()) memo.resize(idx + 1, -1);
        if (memo[idx] < 0) ++memo_len;
        memo[idx] = t;
        return true;
    }
    bool get(size_t idx) { return (idx < memo.size() && memo[idx] >= 0) ? push(memo[idx]) : fail("memo value not found"); }
    bool tuple_of(int k) {
        if (stack.size() < (size_t)k || (!marks.empty() && stack.size() - k < marks.back())) return fail("stack underflow");
        const int32_t id = make(K_TUPLE);
        if (id < 0) return false;
        nodes[id].items.assign(stack.end() - k, stack.end());
        stack.resize(stack.size() - k);
        return push(id);
    }
    bool set_items(int32_t target, const std::vector<int32_t>& kv) {
        if (kv.size() % 2) return fail("odd number of items for SETITEMS");
        Node& t = nodes[target];
        if (t.kind == K_DICT) t.items.insert(t.items.end(), kv.begin(), kv.end());
        else if (t.kind != K_OBJ) return fail("SETITEM on something that is no mapping");
        return true;                                   // (an object built by REDUCE, e.g. an OrderedDict: its items are not looked at)
    }
    bool append(int32_t target, const std::vector<int32_t>& v) {
        Node& t = nodes[target];
        if (t.kind == K_LIST) t.items.insert(t.items.end(), v.begin(), v.end());
        else if (t.kind != K_OBJ) return fail("APPEND on something that is no list");
        return true;
    }

    // Runs the program to its STOP; the result is the node left on the stack.
    bool run(int32_t& result) {
        std::vector<int32_t> tmp;
        for (;;) {
            if (!need(1)) return false;
            const uint8_t op = p[pos++];
            int32_t x, y, z;
            switch (op) {
            case 0x80: if (!need(1)) return false; if (p[pos] > 5) return fail("pickle protocol above 5"); ++pos; break;     // PROTO
            case 0x95: if (!need(8)) return false; pos += 8; break;                                                        // FRAME
            case '.': if (!pop(result)) return false; return true;                                                        // STOP
            case '(': marks.push_back(stack.size()); break;                                                               // MARK
            case '0': if (!pop(x)) return false; break;                                                                   // POP
            case '1': if (!pop_mark(tmp)) return false; break;                                                            // POP_MARK
            case '2': if (!top(x) || !push(x)) return false; break;                                                       // DUP
            case 'N': if (!push(make(K_NONE))) return false; break;
            case 0x88: case 0x89: { const int32_t id = make(K_BOOL); if (id < 0) return false; nodes[id].i = op == 0x88; if (!push(id)) return false; break; }
            case 'J': if (!need(4) || !integer((int32_t)le(4))) return false; break;                                      // BININT
            case 'K': if (!need(1) || !integer((int64_t)le(1))) return false; break;                                      // BININT1
            case 'M': if (!need(2) || !integer((int64_t)le(2))) return false; break;                                      // BININT2
            case 0x8a: case 0x8b: {                                                                                       // LONG1 / LONG4
                const int w = op == 0x8a ? 1 : 4;
                if (!need(w)) return false;
                const uint64_t n = le(w);
                if (n > 8) return fail("integer wider than 64 bits");
                if (!need((int64_t)n)) return false;
                uint64_t v = le((int)n);
                if (n && n < 8 && (v >> (8 * n - 1)) & 1) v |= ~0ull << (8 * n);                                            // sign-extend
                if (!integer((int64_t)v)) return false;
                break;
            }
            case 'G': { if (!need(8)) return false; pos += 8; if (!push(make(K_FLOAT))) return false; break; }            // BINFLOAT (value unused)
            case 0x8c: if (!counted(K_STR, 1)) return false; break;                                                       // SHORT_BINUNICODE
            case 'X': if (!counted(K_STR, 4)) return false; break;                                                        // BINUNICODE
            case 0x8d: if (!counted(K_STR, 8)) return false; break;                                                       // BINUNICODE8
            case 'C': if (!counted(K_BYTES, 1)) return false; break;                                                      // SHORT_BINBYTES
            case 'B': if (!counted(K_BYTES, 4)) return false; break;                                                      // BINBYTES
            case 0x8e: case 0x96: if (!counted(K_BYTES, 8)) return false; break;                                          // BINBYTES8 / BYTEARRAY8
            case ')': if (!push(make(K_TUPLE))) return false; break;
            case 't': { if (!pop_mark(tmp)) return false; const int32_t id = make(K_TUPLE); if (id < 0) return false; nodes[id].items = tmp; if (!push(id)) return false; break; }
            case 0x85: if (!tuple_of(1)) return false; break;
            case 0x86: if (!tuple_of(2)) return false; break;
            case 0x87: if (!tuple_of(3)) return false; break;
            case ']': if (!push(make(K_LIST))) return false; break;
            case '}': if (!push(make(K_DICT))) return false; break;
            case 'a': if (!pop(x) || !top(y)) return false; tmp.assign(1, x); if (!append(y, tmp)) return false; break;   // APPEND
            case 'e': if (!pop_mark(tmp) || !top(y) || !append(y, tmp)) return false; break;                              // APPENDS
            case 's': if (!pop(x) || !pop(y) || !top(z)) return false; tmp = {y, x}; if (!set_items(z, tmp)) return false; break;   // SETITEM
            case 'u': if (!pop_mark(tmp) || !top(z) || !set_items(z, tmp)) return false; break;                           // SETITEMS
            case 'c': {                                                                                                   // GLOBAL: two text lines
                int32_t part[2];
                for (int k = 0; k < 2; ++k) {
                    const void* nl = pos < len ? memchr(p + pos, '\n', (size_t)(len - pos)) : nullptr;
                    if (!nl) return fail("GLOBAL without its newline");
                    part[k] = make(K_STR);
                    if (part[k] < 0) return false;
                    nodes[part[k]].i = pos; nodes[part[k]].n = (const uint8_t*)nl - (p + pos);
                    pos = (const uint8_t*)nl - p + 1;
                }
                const int32_t id = make(K_GLOBAL);
                if (id < 0) return false;
                nodes[id].a = part[0]; nodes[id].b = part[1];
                if (!push(id)) return false;
                break;
            }
            case 0x93: {                                                                                                  // STACK_GLOBAL
                if (!pop(x) || !pop(y)) return false;
                if (nodes[x].kind != K_STR || nodes[y].kind != K_STR) return fail("STACK_GLOBAL wants two strings");
                const int32_t id = make(K_GLOBAL);
                if (id < 0) return false;
                nodes[id].a = y; nodes[id].b = x;
                if (!push(id)) return false;
                break;
            }
            case 'R': case 0x81: {                                                                                        // REDUCE / NEWOBJ: NOT called, only noted
                if (!pop(x) || !pop(y)) return false;
                const int32_t id = make(K_OBJ);
                if (id < 0) return false;
                nodes[id].a = y; nodes[id].b = x;
                if (!push(id)) return false;
                break;
            }
            case 0x92: {                                                                                                  // NEWOBJ_EX
                if (!pop(z) || !pop(x) || !pop(y)) return false;
                const int32_t id = make(K_OBJ);
                if (id < 0) return false;
                nodes[id].a = y; nodes[id].b = x;
                if (!push(id)) return false;
                break;
            }
            case 'b': {                                                                                                   // BUILD
                if (!pop(x) || !top(y)) return false;
                if (nodes[y].kind == K_OBJ) nodes[y].state = x;
                else if (nodes[y].kind != K_DICT) return fail("BUILD on something that is no object");
                break;
            }
            case 'q': if (!need(1) || !put((size_t)le(1))) return false; break;                                           // BINPUT
            case 'r': if (!need(4) || !put((size_t)le(4))) return false; break;                                           // LONG_BINPUT
            case 0x94: if (!put(memo_len)) return false; break;                                                           // MEMOIZE
            case 'h': if (!need(1) || !get((size_t)le(1))) return false; break;                                           // BINGET
            case 'j': if (!need(4) || !get((size_t)le(4))) return false; break;                                           // LONG_BINGET
            default: {
                char b[48];
                snprintf(b, sizeof b, "opcode 0x%02x is outside the subset", op);
                --pos;
                return fail(b);
            }
            }
        }
    }

    bool text_is(int32_t id, const char* s) const {
        if (id < 0 || nodes[id].kind != K_STR) return false;
        const size_t n = strlen(s);
        return (size_t)nodes[id].n == n && memcmp(p + nodes[id].i, s, n) == 0;
    }
    bool global_is(int32_t id, const char* mod_a, const char* mod_b, const char* name) const {
        if (id < 0 || nodes[id].kind != K_GLOBAL) return false;
        return (text_is(nodes[id].a, mod_a) || text_is(nodes[id].a, mod_b)) && text_is(nodes[id].b, name);
    }
    // numpy.dtype('f4' | 'f8') in little-endian or native byte order -> GEM_DT_*, else -1
    int dtype_of(int32_t id) const {
        if (id < 0 || nodes[id].kind != K_OBJ || !global_is(nodes[id].a, "numpy", "numpy", "dtype")) return -1;
        const int32_t args = nodes[id].b;
        if (args < 0 || nodes[args].kind != K_TUPLE || nodes[args].items.empty()) return -1;
        const int32_t code = nodes[args].items[0];
        const int dt = text_is(code, "f4") ? GEM_DT_F32 : text_is(code, "f8") ? GEM_DT_F64 : -1;
        const int32_t st = nodes[id].state;
        if (st >= 0) {                                   // (version, byteorder, ...): '<' little, '=' native, '|' not applicable
            if (nodes[st].kind != K_TUPLE || nodes[st].items.size() < 2) return -1;
            const int32_t bo = nodes[st].items[1];
            if (!(text_is(bo, "<") || text_is(bo, "=") || text_is(bo, "|"))) return -1;
        }
        return dt;
    }
    bool shape_of(int32_t id, gem_pickle_array& out) const {
        if (id < 0 || nodes[id].kind != K_TUPLE || nodes[id].items.size() > 4) return false;
        out.ndim = (int32_t)nodes[id].items.size();
        for (int k = 0; k < 4; ++k) out.shape[k] = 1;
        for (int k = 0; k < out.ndim; ++k) {
            const Node& d = nodes[nodes[id].items[k]];
            if (d.kind != K_INT || d.i < 0 || d.i > (1ll << 40)) return false;
            out.shape[k] = d.i;
        }
        return true;
    }
    // One numpy.ndarray of float32 / float64, as numpy's __reduce__ (protocols 2-4) or __reduce_ex__(5) describes it
    bool array_of(int32_t id, gem_pickle_array& out) const {
        if (id < 0 || nodes[id].kind != K_OBJ) return false;
        const Node& o = nodes[id];
        int32_t shape = -1, dtype = -1, raw = -1;
        int fortran = 0;
        if (global_is(o.a, "numpy.core.multiarray", "numpy._core.multiarray", "_reconstruct")) {
            // state = (version, shape, dtype, is_fortran, rawdata)
            if (o.state < 0 || nodes[o.state].kind != K_TUPLE || nodes[o.state].items.size() != 5) return false;
            const std::vector<int32_t>& s = nodes[o.state].items;
            if (nodes[s[0]].kind != K_INT || nodes[s[3]].kind != K_BOOL) return false;
            shape = s[1]; dtype = s[2]; fortran = (int)nodes[s[3]].i; raw = s[4];
        } else if (global_is(o.a, "numpy.core.numeric", "numpy._core.numeric", "_frombuffer")) {
            // arguments = (buffer, dtype, shape, order)
            if (o.b < 0 || nodes[o.b].kind != K_TUPLE || nodes[o.b].items.size() != 4 || o.state >= 0) return false;
            const std::vector<int32_t>& s = nodes[o.b].items;
            raw = s[0]; dtype = s[1]; shape = s[2];
            if (text_is(s[3], "F")) fortran = 1;
            else if (!text_is(s[3], "C")) return false;
        } else {
            return false;
        }
        if (raw < 0 || nodes[raw].kind != K_BYTES) return false;         // (protocol 2 writes the bytes as latin-1 text: not this path)
        const int dt = dtype_of(dtype);
        if (dt < 0 || !shape_of(shape, out)) return false;
        int64_t count = 1;
        for (int k = 0; k < out.ndim; ++k) {
            if (out.shape[k] && count > (1ll << 40) / out.shape[k]) return false;
            count *= out.shape[k];
        }
        const int64_t item = dt == GEM_DT_F32 ? 4 : 8;
        if (nodes[raw].n != count * item) return false;
        out.offset = nodes[raw].i; out.nbytes = nodes[raw].n; out.dtype = dt;
        out.fortran = (out.ndim > 1 && fortran) ? 1 : 0;
        return true;
    }
};

// gem_pickle_scan's work: the arrays under keys[k], key by key, and their number per key (-1: no such key)
int scan_image(const void* h_image, int64_t len, const char* const* keys, int n_keys, std::vector<gem_pickle_array>& out,
               std::vector<int64_t>& counts) {
    if (!h_image || len < 2 || !keys || n_keys < 1 || n_keys > 64) { set_error("gem_pickle_scan: bad argument"); return 1; }
    Scanner s;
    s.p = static_cast<const uint8_t*>(h_image);
    s.len = len;
    int32_t root = -1;
    out.clear();
    counts.assign(n_keys, -1);
    try {
        s.nodes.reserve(8192);
        if (!s.run(root)) { set_error("gem_pickle_scan: " + s.why); return GEM_PICKLE_UNSUPPORTED; }
        if (s.nodes[root].kind != K_DICT) { set_error("gem_pickle_scan: the pickle does not hold a dict"); return GEM_PICKLE_UNSUPPORTED; }
        const std::vector<int32_t>& kv = s.nodes[root].items;
        for (int k = 0; k < n_keys; ++k) {
            int32_t val = -1;
            for (size_t i = 0; i + 1 < kv.size(); i += 2)
                if (s.text_is(kv[i], keys[k])) val = kv[i + 1];          // (a repeated key: the last one stands, as in a dict)
            if (val < 0) continue;                                       // the caller raises KeyError like the reference (optimizer.py:318-324)
            const Node& v = s.nodes[val];
            if (v.kind != K_LIST && v.kind != K_TUPLE) { set_error(std::string("gem_pickle_scan: '") + keys[k] + "' is not a list of arrays"); return GEM_PICKLE_UNSUPPORTED; }
            counts[k] = (int64_t)v.items.size();
            for (int32_t el : v.items) {
                gem_pickle_array a;
                memset(&a, 0, sizeof a);
                if (!s.array_of(el, a)) {
                    set_error(std::string("gem_pickle_scan: an element of '") + keys[k] + "' is not a little-endian float32 / float64 ndarray");
                    return GEM_PICKLE_UNSUPPORTED;
                }
                a.key = k;
                out.push_back(a);
            }
        }
    } catch (const std::bad_alloc&) {
        set_error("gem_pickle_scan: out of memory");
        return 1;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------ the device side
struct HeatArgs {
    const uint32_t* image;          // device image of the file (4-byte aligned base), image_len bytes of it valid
    const int64_t* offsets;         // byte offset of every frame's payload in the image
    float* out;                     // [n][H][W][J]
    int64_t image_len;
    int H, W, J, TH;
};

// the 32-bit word that starts at BYTE address `at` of the image, whatever `at`'s alignment (words past the end read as 0)
__device__ __forceinline__ uint32_t word_at(const HeatArgs& a, int64_t at) {
    const int64_t k = at >> 2;
    if (k < 0) return 0u;                                    // (an offset in front of the image: the caller's table is wrong, nothing is read)
    const uint32_t lo = 4 * k + 4 <= a.image_len ? a.image[k] : 0u;
    const int m = (int)(at & 3);
    if (m == 0) return lo;
    const uint32_t hi = 4 * k + 8 <= a.image_len ? a.image[k + 1] : 0u;
    return __builtin_amdgcn_alignbyte(hi, lo, m);
}

template <bool F64>
__device__ __forceinline__ float element_at(const HeatArgs& a, int64_t payload, int64_t e) {
    if (F64) {
        const int64_t at = payload + 8 * e;
        const uint64_t bits = (uint64_t)word_at(a, at) | ((uint64_t)word_at(a, at + 4) << 32);
        return (float)__longlong_as_double((long long)bits);          // round to nearest even, as ndarray.astype / Tensor.float do
    }
    return __uint_as_float(word_at(a, payload + 4 * e));
}

// C-ordered payloads: frame f's [H][W][J] elements are a straight (cast) copy.  grid (ceil(HWJ / 1024), n), 256 threads x 4.
template <bool F64>
__global__ __launch_bounds__(256) void heat_copy_kernel(HeatArgs a) {
    const int f = blockIdx.y;
    const int64_t per = (int64_t)a.H * a.W * a.J, payload = a.offsets[f];
    const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    float* dst = a.out + (int64_t)f * per;
    if (e0 + 4 <= per && (per & 3) == 0) {
        float4 v;
        v.x = element_at<F64>(a, payload, e0);     v.y = element_at<F64>(a, payload, e0 + 1);
        v.z = element_at<F64>(a, payload, e0 + 2); v.w = element_at<F64>(a, payload, e0 + 3);
        *reinterpret_cast<float4*>(dst + e0) = v;
    } else {
        for (int64_t e = e0; e < e0 + 4 && e < per; ++e) dst[e] = element_at<F64>(a, payload, e);
    }
}

// Fortran-ordered payloads (what loadmat returns): the payload is the C-ordered [J][W][H] image, out wants [H][W][J].
// One workgroup moves WT columns w of one frame (all h, all j) through LDS: for every joint the WT * H elements of those columns are
// ONE contiguous run of the payload (4 KB for 16 columns of 64 rows: whole cache lines, consecutive lanes on consecutive words); on
// the way out every row h is one contiguous run of WT * J floats (960 B; the launcher takes WT = 8: 2 KB in, 480 B out).  LDS pitches
// H + 1 per (j, w) run and WT * (H + 1) + 5 per joint keep both sides at two-way bank conflicts at most.  HC / WC / JC / WTC: the geometry as constants (0 = from the arguments;
// the 64 x 64 x 15 maps of the path divide by constants only).
constexpr int TR_PAD_J = 5;
template <bool F64, int HC, int WC, int JC, int WTC>
__global__ __launch_bounds__(256) void heat_transpose_kernel(HeatArgs a) {
    extern __shared__ float tile[];
    const int H = HC ? HC : a.H, W = WC ? WC : a.W, J = JC ? JC : a.J, WT = WTC ? WTC : a.TH;
    const int f = blockIdx.y, w0 = blockIdx.x * WT, wn = WTC ? WTC : min(WT, W - w0);
    const int PH = H + 1, PJ = WT * PH + TR_PAD_J;
    const int64_t payload = a.offsets[f];
    const int run = wn * H;                           // elements of one joint's run
#pragma unroll 4
    for (int r = threadIdx.x; r < J * run; r += 256) {
        const int j = r / run, rem = r - j * run, w = rem / H, h = rem - w * H;
        tile[j * PJ + w * PH + h] = element_at<F64>(a, payload, ((int64_t)j * W + w0) * H + rem);
    }
    __syncthreads();
    float* dst = a.out + ((int64_t)f * H * W + w0) * J;
    const int orun = wn * J;                          // floats of one output row's run
#pragma unroll 4
    for (int r = threadIdx.x; r < H * orun; r += 256) {
        const int h = r / orun, rem = r - h * orun, w = rem / J, j = rem - w * J;
        dst[(int64_t)h * W * J + rem] = tile[j * PJ + w * PH + h];
    }
}

}  // namespace
}  // namespace gem

using namespace gem;

extern "C" {

int gem_pickle_scan(const void* h_image, int64_t len, const char* const* keys, int n_keys, gem_pickle_array* out, int64_t cap,
                    int64_t* counts) {
    if (!out || cap < 0 || !counts) { set_error("gem_pickle_scan: bad argument"); return 1; }
    std::vector<gem_pickle_array> found;
    std::vector<int64_t> cnt;
    const int rc = scan_image(h_image, len, keys, n_keys, found, cnt);
    if (rc) return rc;
    if ((int64_t)found.size() > cap) { set_error("gem_pickle_scan: more arrays than the caller made room for"); return 1; }
    if (!found.empty()) memcpy(out, found.data(), found.size() * sizeof(gem_pickle_array));
    for (int k = 0; k < n_keys; ++k) counts[k] = cnt[k];
    return 0;
}

int gem_pickle_gather_f64(const void* h_image, int64_t len, const gem_pickle_array* arrays, int64_t n, double* h_out) {
    if (!h_image || !arrays || n < 0 || (n && !h_out)) { set_error("gem_pickle_gather_f64: bad argument"); return 1; }
    const uint8_t* p = static_cast<const uint8_t*>(h_image);
    for (int64_t i = 0; i < n; ++i) {
        const gem_pickle_array& a = arrays[i];
        const gem_pickle_array& a0 = arrays[0];
        const int64_t item = a.dtype == GEM_DT_F32 ? 4 : 8;
        int64_t count = 1;
        for (int k = 0; k < 4; ++k) count *= a.shape[k];
        if ((a.dtype != GEM_DT_F32 && a.dtype != GEM_DT_F64) || a.ndim < 0 || a.ndim > 4 || a.offset < 0 || a.nbytes != count * item || a.offset + a.nbytes > len) {
            set_error("gem_pickle_gather_f64: an array lies outside the image"); return 1;
        }
        if (a.ndim != a0.ndim || memcmp(a.shape, a0.shape, sizeof a.shape)) { set_error("gem_pickle_gather_f64: the arrays differ in shape"); return GEM_PICKLE_UNSUPPORTED; }
        const uint8_t* src = p + a.offset;
        double* dst = h_out + i * count;
        const int64_t s0 = a.shape[0], s1 = a.shape[1], s2 = a.shape[2], s3 = a.shape[3];
        for (int64_t c = 0; c < count; ++c) {
            int64_t e = c;
            if (a.fortran) {                               // C index (i0,i1,i2,i3) -> position in the Fortran-ordered payload
                const int64_t i3 = c % s3, i2 = (c / s3) % s2, i1 = (c / (s3 * s2)) % s1, i0 = c / (s3 * s2 * s1);
                e = i0 + s0 * (i1 + s1 * (i2 + s2 * i3));
            }
            if (a.dtype == GEM_DT_F32) { float v; memcpy(&v, src + 4 * e, 4); dst[c] = (double)v; }
            else memcpy(dst + c, src + 8 * e, 8);
        }
    }
    return 0;
}

int gem_heat_gather(const void* d_image, int64_t image_len, const int64_t* d_offsets, int64_t n, int heat_h, int heat_w, int n_joints,
                    int dtype, int fortran, float* d_out, void* stream) {
    if (n == 0) return 0;
    if (!d_image || !d_offsets || !d_out || n < 0 || n > 65535 || heat_h < 1 || heat_w < 1 || n_joints < 1 || image_len < 4) {
        set_error("gem_heat_gather: bad argument (at most 65535 frames per call)"); return 1;
    }
    if (dtype != GEM_DT_F32 && dtype != GEM_DT_F64) { set_error("gem_heat_gather: dtype must be GEM_DT_F32 or GEM_DT_F64"); return 1; }
    if (reinterpret_cast<uintptr_t>(d_image) & 3) { set_error("gem_heat_gather: the image must start on a 4-byte boundary"); return 1; }
    if (reinterpret_cast<uintptr_t>(d_out) & 15) { set_error("gem_heat_gather: the output must start on a 16-byte boundary"); return 1; }
    HeatArgs a;
    a.image = static_cast<const uint32_t*>(d_image); a.offsets = d_offsets; a.out = d_out; a.image_len = (image_len + 3) & ~3ll;          // (whole words: the allocation is readable to there, see gem_hip.h)
    a.H = heat_h; a.W = heat_w; a.J = n_joints; a.TH = 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t per = (int64_t)heat_h * heat_w * n_joints;
    if (per > (1ll << 30)) { set_error("gem_heat_gather: heat-maps too large"); return 1; }
    if (!fortran) {
        const dim3 grid((unsigned)((per + 1023) / 1024), (unsigned)n);
        if (dtype == GEM_DT_F64) hipLaunchKernelGGL(heat_copy_kernel<true>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(heat_copy_kernel<false>, grid, dim3(256), 0, s, a);
    } else {
        // columns per workgroup: 8 (31 KB of LDS at 64 x 64 x 15: five workgroups per CU hide the loads' latency; 16 columns -- two
        // workgroups per CU -- measured 0.71 against 0.34 ms for 2000 frames, 2.9 TB/s), fewer where the tile would not fit 64 KB
        int WT = 8;
        auto lds_of = [&](int wt) { return (size_t)n_joints * ((size_t)wt * (heat_h + 1) + TR_PAD_J) * 4; };
        while (WT > 1 && (WT > heat_w || lds_of(WT) > 64 * 1024)) WT >>= 1;
        if (lds_of(WT) > 64 * 1024) { set_error("gem_heat_gather: a heat-map column of H * J floats does not fit the transposing tile"); return 1; }
        a.TH = WT;
        const dim3 grid((unsigned)((heat_w + WT - 1) / WT), (unsigned)n);
        const size_t lds = lds_of(WT);
        if (heat_h == 64 && heat_w == 64 && n_joints == 15) {
            if (dtype == GEM_DT_F64) hipLaunchKernelGGL((heat_transpose_kernel<true, 64, 64, 15, 8>), grid, dim3(256), lds, s, a);
            else hipLaunchKernelGGL((heat_transpose_kernel<false, 64, 64, 15, 8>), grid, dim3(256), lds, s, a);
        } else {
            if (dtype == GEM_DT_F64) hipLaunchKernelGGL((heat_transpose_kernel<true, 0, 0, 0, 0>), grid, dim3(256), lds, s, a);
            else hipLaunchKernelGGL((heat_transpose_kernel<false, 0, 0, 0, 0>), grid, dim3(256), lds, s, a);
        }
    }
    GEM_HIP(hipGetLastError());
    return 0;
}

struct gem_chunk {
    int fd = -1;
    const uint8_t* map = nullptr;
    int64_t len = 0;
    std::vector<gem_pickle_array> arrays;      // key by key
    std::vector<int64_t> counts, first;        // per key: number of arrays (-1: no such key), index of its first array
};

void gem_chunk_close(gem_chunk* c) {
    if (!c) return;
    if (c->map) munmap(const_cast<uint8_t*>(c->map), (size_t)c->len);
    if (c->fd >= 0) close(c->fd);
    delete c;
}

int gem_chunk_open(const char* path, const char* const* keys, int n_keys, gem_chunk** out) {
    if (!path || !out) { set_error("gem_chunk_open: bad argument"); return 1; }
    *out = nullptr;
    gem_chunk* c = new (std::nothrow) gem_chunk;
    if (!c) { set_error("gem_chunk_open: out of memory"); return 1; }
    c->fd = open(path, O_RDONLY | O_CLOEXEC);
    struct stat st;
    if (c->fd < 0 || fstat(c->fd, &st) != 0 || st.st_size < 2) {
        set_error(std::string("gem_chunk_open: ") + path + ": " + (c->fd < 0 ? strerror(errno) : "not a pickle"));
        gem_chunk_close(c);
        return 1;
    }
    c->len = (int64_t)st.st_size;
    void* m = mmap(nullptr, (size_t)c->len, PROT_READ, MAP_PRIVATE, c->fd, 0);
    if (m == MAP_FAILED) { set_error(std::string("gem_chunk_open: mmap: ") + strerror(errno)); gem_chunk_close(c); return 1; }
    c->map = static_cast<const uint8_t*>(m);
    const int rc = scan_image(c->map, c->len, keys, n_keys, c->arrays, c->counts);
    if (rc) { gem_chunk_close(c); return rc; }
    c->first.assign(n_keys, 0);
    int64_t at = 0;
    for (int k = 0; k < n_keys; ++k) { c->first[k] = at; at += c->counts[k] > 0 ? c->counts[k] : 0; }
    *out = c;
    return 0;
}

int64_t gem_chunk_bytes(const gem_chunk* c) { return c ? c->len : -1; }

int64_t gem_chunk_count(const gem_chunk* c, int key) { return (c && key >= 0 && key < (int)c->counts.size()) ? c->counts[key] : -1; }

int gem_chunk_info(const gem_chunk* c, int key, int64_t* info) {
    const int64_t n = gem_chunk_count(c, key);
    if (!c || key < 0 || key >= (int)c->counts.size() || !info) { set_error("gem_chunk_info: bad argument"); return 1; }
    for (int k = 0; k < 8; ++k) info[k] = 0;
    info[0] = n;
    if (n <= 0) return 0;
    const gem_pickle_array* a = c->arrays.data() + c->first[key];
    info[1] = a[0].ndim; info[2] = a[0].dtype; info[3] = a[0].fortran;
    for (int k = 0; k < 4; ++k) info[4 + k] = a[0].shape[k];
    for (int64_t i = 1; i < n; ++i)
        if (a[i].ndim != a[0].ndim || a[i].dtype != a[0].dtype || a[i].fortran != a[0].fortran || memcmp(a[i].shape, a[0].shape, sizeof a[0].shape)) {
            info[1] = -1;                             // not uniform: the caller un-pickles this entry the ordinary way
            break;
        }
    return 0;
}

int gem_chunk_offsets(const gem_chunk* c, int key, int64_t* h_out, int64_t cap) {
    const int64_t n = gem_chunk_count(c, key);
    if (n < 0 || !h_out || cap < n) { set_error("gem_chunk_offsets: no such key, or too little room"); return 1; }
    for (int64_t i = 0; i < n; ++i) h_out[i] = c->arrays[c->first[key] + i].offset;
    return 0;
}

int gem_chunk_gather_f64(const gem_chunk* c, int key, double* h_out, int64_t n_out) {
    const int64_t n = gem_chunk_count(c, key);
    if (n < 0) { set_error("gem_chunk_gather_f64: no such key"); return 1; }
    if (n == 0) return 0;
    const gem_pickle_array& a0 = c->arrays[c->first[key]];
    int64_t each = 1;
    for (int k = 0; k < 4; ++k) each *= a0.shape[k];
    if (n_out < n * each) { set_error("gem_chunk_gather_f64: too little room"); return 1; }
    return gem_pickle_gather_f64(c->map, c->len, c->arrays.data() + c->first[key], n, h_out);
}

int gem_file_stage(const char* path, int device, void* h_pinned, void* d_image, int64_t buffer_bytes, int64_t slice_bytes,
                   int64_t* file_bytes, void* stream) {
    if (!path || !h_pinned || !d_image || !file_bytes) { set_error("gem_file_stage: bad argument"); return 1; }
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) {
        set_error(std::string("gem_file_stage: ") + path + ": " + strerror(errno));
        if (fd >= 0) close(fd);
        return 1;
    }
    const int64_t len = (int64_t)st.st_size;
    *file_bytes = len;
    int rc = 0;
    if (len + 8 > buffer_bytes) { set_error("gem_file_stage: the staging buffers are too small for this file"); rc = 1; }
    if (slice_bytes < 65536) slice_bytes = 65536;
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint8_t* host = static_cast<uint8_t*>(h_pinned);
    uint8_t* image = static_cast<uint8_t*>(d_image);
    if (!rc && !hip_ok(hipSetDevice(device), "hipSetDevice")) rc = 1;
    for (int64_t o = 0; !rc && o < len; o += slice_bytes) {
        const int64_t want = len - o < slice_bytes ? len - o : slice_bytes;
        for (int64_t got = 0; got < want;) {
            const ssize_t r = pread(fd, host + o + got, (size_t)(want - got), (off_t)(o + got));
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) { set_error(std::string("gem_file_stage: short read: ") + (r < 0 ? strerror(errno) : "the file shrank")); rc = 1; break; }
            got += r;
        }
        // (the copy takes whole words: the slice's last word may reach a few bytes past the file, inside the buffers)
        if (!rc && !hip_ok(hipMemcpyAsync(image + o, host + o, (size_t)((want + 3) & ~3ll), hipMemcpyHostToDevice, s), "hipMemcpyAsync")) rc = 1;
    }
    close(fd);
    return rc;
}

}  // extern "C"
