// Flat parameter layout + error plumbing + time-feature tables (host side).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.hpp"

namespace mctsq {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

namespace {
struct Cursor {
    NetLayout* L;
    int64_t pos = 0;
    int64_t add(int64_t n) {
        int64_t at = pos;
        if (L->n_tensors < 160) {
            L->t_off[L->n_tensors] = at;
            L->t_size[L->n_tensors] = n;
        }
        L->n_tensors++;
        pos = align16(at + n);
        return at;
    }
};
}  // namespace

// Order = torch named_parameters() of TripleEncoderDQN (dqn_per.py:178-214): direct Parameters
// of a module first (positional_encoding), then its sub-modules in registration order.
int build_layout(const mctsq_net_desc* net, NetLayout* L) {
    MCTSQ_CHECK(net != nullptr, "net descriptor is NULL");
    MCTSQ_CHECK(net->embed_dim == E, "embed_dim must be %d (got %d)", E, net->embed_dim);
    MCTSQ_CHECK(net->hidden_layers >= 1 && net->hidden_layers <= 4, "hidden_layers must be 1..4");
    MCTSQ_CHECK(net->hidden_units % 16 == 0 && net->hidden_units >= 16 && net->hidden_units <= 256,
                "hidden_units must be a multiple of 16 in 16..256 (got %d)", net->hidden_units);
    MCTSQ_CHECK(net->action_dim >= 1 && net->action_dim <= 16, "action_dim must be 1..16");
    MCTSQ_CHECK(net->activation == MCTSQ_ACT_RELU || net->activation == MCTSQ_ACT_TANH,
                "activation must be relu(0) or tanh(1)");
    memset(L, 0, sizeof(*L));
    Cursor c{L};
    for (int s = 0; s < 3; ++s) {
        const mctsq_encoder_desc& d = net->enc[s];
        EncOff& e = L->enc[s];
        e.kind = d.kind;
        e.S = d.seq_len;
        e.Fd = d.data_dim;
        e.tf = d.time_feat;
        e.F = d.data_dim + (d.time_feat != MCTSQ_TF_NONE ? 2 : 0);
        MCTSQ_CHECK(e.S >= 1 && e.S <= MAX_SEQ, "encoder %d: seq_len %d outside 1..%d", s, e.S, MAX_SEQ);
        MCTSQ_CHECK(e.Fd >= 1 && e.F <= MAX_F, "encoder %d: feature width %d outside 1..%d", s, e.F, MAX_F);
        MCTSQ_CHECK(d.time_feat >= 0 && d.time_feat <= 2, "encoder %d: bad time_feat", s);
        MCTSQ_CHECK(d.time_feat != MCTSQ_TF_GAS_HOURS || e.S == 3, "gas hour features need seq_len 3");
        const int64_t F = e.F;
        switch (d.kind) {
        case MCTSQ_ENC_GRU:   // dqn_per.py:62-64
            e.gru.wih = c.add(G3 * F); e.gru.whh = c.add(G3 * E);
            e.gru.bih = c.add(G3); e.gru.bhh = c.add(G3);
            e.gru.aw = c.add(E); e.gru.ab = c.add(1);
            e.gru.lnw = c.add(E); e.gru.lnb = c.add(E);
            break;
        case MCTSQ_ENC_CONV:  // dqn_per.py:25-37
            e.conv.w1 = c.add(E * F * 3); e.conv.b1 = c.add(E);
            e.conv.w2 = c.add(E * E * 3); e.conv.b2 = c.add(E);
            e.conv.aw = c.add(E); e.conv.ab = c.add(1);
            e.conv.lnw = c.add(E); e.conv.lnb = c.add(E);
            break;
        case MCTSQ_ENC_TRANSFORMER:  // dqn_per.py:84-88, torch TransformerEncoderLayer defaults
            e.tfm.pos = c.add(500 * E);
            e.tfm.ew = c.add(E * F); e.tfm.eb = c.add(E);
            for (int l = 0; l < 2; ++l) {
                TfLayerOff& t = e.tfm.layer[l];
                t.inw = c.add(G3 * E); t.inb = c.add(G3);
                t.ow = c.add(E * E); t.ob = c.add(E);
                t.l1w = c.add(2048 * E); t.l1b = c.add(2048);
                t.l2w = c.add(E * 2048); t.l2b = c.add(E);
                t.n1w = c.add(E); t.n1b = c.add(E);
                t.n2w = c.add(E); t.n2b = c.add(E);
            }
            e.tfm.lnw = c.add(E); e.tfm.lnb = c.add(E);
            break;
        case MCTSQ_ENC_MLP:   // dqn_per.py:109-114 (gas/EUA slot only, sized input_dim*3)
            MCTSQ_CHECK(s == 2, "mlp encoder is only defined for the gas/EUA slot");
            MCTSQ_CHECK(e.S == 3, "gas/EUA mlp encoder needs seq_len 3 (Linear(input_dim*3))");
            e.mlp.w0 = c.add(E * 3 * F); e.mlp.b0 = c.add(E);
            e.mlp.w1 = c.add(E * E); e.mlp.b1 = c.add(E);
            break;
        default:
            MCTSQ_CHECK(false, "encoder %d: unknown kind %d", s, d.kind);
        }
        MCTSQ_CHECK(!(s == 2 && (d.kind == MCTSQ_ENC_CONV || d.kind == MCTSQ_ENC_TRANSFORMER)),
                    "gas/EUA encoder must be mlp or gru (dqn_per.py:108-120)");
    }
    L->n_fc = net->hidden_layers + 1;
    L->act = net->activation;
    L->A = net->action_dim;
    L->H = net->hidden_units;
    int in = FEAT;
    for (int i = 0; i < L->n_fc; ++i) {
        int out = (i == L->n_fc - 1) ? net->action_dim : net->hidden_units;
        L->fc_in[i] = in; L->fc_out[i] = out;
        L->fc_w[i] = c.add((int64_t)out * in);
        L->fc_b[i] = c.add(out);
        in = out;
    }
    L->total = c.pos;
    MCTSQ_CHECK(L->n_tensors <= 160, "too many parameter tensors");
    return 0;
}

// ---- time features --------------------------------------------------------------------
// mctsq_config_dqn.py:563-567 with the defaults every caller uses (step 15 min, start 0: a
// period-4 pattern) and :582-585 with hours [-12, 0, 12]; the reference computes them in fp64 and
// casts to fp32 (torch.FloatTensor), so the tiny non-zero cos(pi/2) etc. are kept.
void host_time_features(int mode, int S, float* out) {
    const double PI = 3.14159265358979323846;
    for (int t = 0; t < S; ++t) {
        double rad;
        if (mode == MCTSQ_TF_MINUTE) {
            rad = 2 * PI * (double)((t * 15) % 60) / 60;
        } else {
            const int hours[3] = {12, 0, 12};   // (-12 % 24, 0, 12 % 24) with Python's modulo
            rad = 2 * PI * (double)hours[t % 3] / 24;
        }
        out[2 * t] = (float)sin(rad);
        out[2 * t + 1] = (float)cos(rad);
    }
}

}  // namespace mctsq
