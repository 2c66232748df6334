// Shared declarations for libmctsq_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mctsq_hip.h"

namespace mctsq {

constexpr int E = 64;          // embed_dim (dqn_per.py: embed_dim, config_mctsq.yaml:37)
constexpr int G3 = 3 * E;      // GRU gate rows r|z|n
constexpr int TILE = 16;       // samples per workgroup tile = MFMA 16x16x4 M dimension
constexpr int FEAT = 3 * E;    // concatenated feature width (dqn_per.py:235)
constexpr int MAX_SEQ = 32;    // longest window handled by the fused encoders (price: 25)
constexpr int MAX_F = 8;       // widest per-step feature vector (process 6 + 2 time features)
constexpr float LN_EPS = 1e-5f;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void set_error(const char* fmt, ...);
#define MCTSQ_CHECK(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            mctsq::set_error(__VA_ARGS__); \
            return -1;                    \
        }                                 \
    } while (0)
#define MCTSQ_HIP(call)                                                            \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                    \
            mctsq::set_error("%s failed: %s", #call, hipGetErrorString(e_));       \
            return -2;                                                             \
        }                                                                          \
    } while (0)

inline int64_t align16(int64_t n) { return (n + 15) & ~int64_t(15); }
inline int64_t round_up(int64_t n, int64_t m) { return (n + m - 1) / m * m; }

// ---- parameter offsets (floats) inside the flat buffer, torch named_parameters() order ----
struct GruOff { int64_t wih, whh, bih, bhh, aw, ab, lnw, lnb; };
struct ConvOff { int64_t w1, b1, w2, b2, aw, ab, lnw, lnb; };
struct MlpOff { int64_t w0, b0, w1, b1; };
struct TfLayerOff { int64_t inw, inb, ow, ob, l1w, l1b, l2w, l2b, n1w, n1b, n2w, n2b; };
struct TfOff { int64_t pos, ew, eb; TfLayerOff layer[2]; int64_t lnw, lnb; };
struct EncOff {
    int kind, S, Fd, F, tf;     // F = Fd + 2 when time features are appended
    GruOff gru; ConvOff conv; MlpOff mlp; TfOff tfm;
};
struct NetLayout {
    EncOff enc[3];
    int n_fc;                   // hidden_layers + 1 linears
    int fc_in[5], fc_out[5];
    int64_t fc_w[5], fc_b[5];
    int act, A, H;
    int64_t total;              // padded float count
    int n_tensors;
    int64_t t_off[160], t_size[160];
};
int build_layout(const mctsq_net_desc* net, NetLayout* L);   // <0 on error (message set)

// time features of the non-_per flavour: host table (for tests) and the device-side literal form
void host_time_features(int mode, int S, float* out);
const char* last_error();

}  // namespace mctsq
