// Host-side helpers: error text, network spec derivation.
#include <math.h>
#include <stdarg.h>

#include <mutex>
#include <string>
#include <vector>

#include "lnr_common.h"

static thread_local char g_err[512] = "";

void lnr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* lnr_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// Optional per-kernel timing of the entry points that launch several kernels (lnr_density_forward / _backward):
// HIP events recorded on the caller's stream around every internal launch.  Off by default (no events, no cost).
// ------------------------------------------------------------------------------------------------
namespace {
struct ProfSpan { int name; hipEvent_t start, stop; };
std::mutex g_prof_mutex;
bool g_prof_on = false;
std::vector<std::string> g_prof_names;
std::vector<ProfSpan> g_prof_spans;
std::vector<hipEvent_t> g_prof_pool;

hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

bool lnr_profile_active() { return g_prof_on; }

int lnr_profile_begin(const char* name, hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (!g_prof_on) return -1;
    int id = -1;
    for (size_t i = 0; i < g_prof_names.size(); ++i) if (g_prof_names[i] == name) { id = (int)i; break; }
    if (id < 0) { g_prof_names.push_back(name); id = (int)g_prof_names.size() - 1; }
    ProfSpan sp; sp.name = id; sp.start = prof_event(); sp.stop = prof_event();
    if (!sp.start || !sp.stop) return -1;
    hipEventRecord(sp.start, st);
    g_prof_spans.push_back(sp);
    return (int)g_prof_spans.size() - 1;
}

void lnr_profile_end(int span, hipStream_t st) {
    if (span < 0) return;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (span < (int)g_prof_spans.size()) hipEventRecord(g_prof_spans[span].stop, st);
}

extern "C" int lnr_profile_enable(int32_t on) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_on = on != 0;
    // events are created here, not inside the timed launches (a few thousand cover ~100 iterations between two reads); a caller that
    // toggles profiling per iteration (bench.py samples every 4th) must not pay for event creation each time: refill only when low
    if (g_prof_on && g_prof_pool.size() >= 256) return LNR_OK;
    while (g_prof_on && g_prof_pool.size() < 4096) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) break;
        g_prof_pool.push_back(e);
    }
    return LNR_OK;
}

extern "C" int lnr_profile_read(char* names, int32_t name_stride, float* total_ms, int32_t* calls, int32_t capacity) {
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    std::vector<double> ms(g_prof_names.size(), 0.0);
    std::vector<int> n(g_prof_names.size(), 0);
    for (const ProfSpan& sp : g_prof_spans) {
        float t = 0.0f;
        if (hipEventSynchronize(sp.stop) == hipSuccess && hipEventElapsedTime(&t, sp.start, sp.stop) == hipSuccess) { ms[sp.name] += t; n[sp.name]++; }
        g_prof_pool.push_back(sp.start); g_prof_pool.push_back(sp.stop);
    }
    g_prof_spans.clear();
    int out = 0;
    for (size_t i = 0; i < g_prof_names.size() && out < capacity; ++i) {
        if (n[i] == 0) continue;
        if (names && name_stride > 0) { strncpy(names + (size_t)out * name_stride, g_prof_names[i].c_str(), name_stride - 1); names[(size_t)out * name_stride + name_stride - 1] = 0; }
        if (total_ms) total_ms[out] = (float)ms[i];
        if (calls) calls[out] = n[i];
        ++out;
    }
    return out;
}
extern "C" int lnr_version(void) { return 200; }

// Level geometry of the multiresolution grid, as published for tiny-cuda-nn's GridEncoding:
//   scale = 2^(level*log2(per_level_scale)) * base - 1 ; res = ceil(scale) + 1
//   size  = min(roundup8(res^3), 2^log2_table) ; hashed iff res^3 > size
extern "C" int lnr_net_spec_finalize(LnrNetSpec* s) {
    LNR_REQUIRE(s != nullptr, "lnr_net_spec_finalize: null spec");
    LNR_REQUIRE(s->n_neurons == 16 || s->n_neurons == 32 || s->n_neurons == 64 || s->n_neurons == 128 || s->n_neurons == 256,
                "n_neurons must be 16, 32, 64, 128 or 256, got %d", s->n_neurons);
    LNR_REQUIRE(s->n_hidden >= 1 && s->n_hidden <= 8, "n_hidden_layers must be in [1,8], got %d", s->n_hidden);
    LNR_REQUIRE(s->activation >= LNR_ACT_NONE && s->activation <= LNR_ACT_TANH, "unknown activation %d", s->activation);
    LNR_REQUIRE(s->precision == LNR_PREC_F32 || s->precision == LNR_PREC_F16 || s->precision == LNR_PREC_F32_CHAIN, "unknown precision %d", s->precision);
    LNR_REQUIRE(s->pos_rounding == LNR_POS_FMA || s->pos_rounding == LNR_POS_MUL_ADD, "unknown pos_rounding %d", s->pos_rounding);
    if (s->encoding == LNR_ENC_HASHGRID) {
        LNR_REQUIRE(s->n_levels >= 1 && s->n_levels <= LNR_MAX_LEVELS, "n_levels must be in [1,%d]", LNR_MAX_LEVELS);
        LNR_REQUIRE(s->n_features == 1 || s->n_features == 2 || s->n_features == 4 || s->n_features == 8,
                    "n_features_per_level must be 1, 2, 4 or 8, got %d", s->n_features);
        LNR_REQUIRE(s->log2_table >= 4 && s->log2_table <= 28, "log2_hashmap_size out of range: %d", s->log2_table);
        LNR_REQUIRE(s->base_res >= 1, "base_resolution must be >= 1");
        const uint64_t cap = 1ull << s->log2_table;
        uint64_t off = 0;
        const float l2 = log2f(s->per_level_scale);
        for (int l = 0; l < s->n_levels; ++l) {
            float scale = exp2f((float)l * l2) * (float)s->base_res - 1.0f;
            uint32_t res = (uint32_t)ceilf(scale) + 1u;
            uint64_t dense = (uint64_t)res * res * res;
            uint64_t size = ((dense + 7ull) / 8ull) * 8ull;
            if (size > cap) size = cap;
            s->level_scale[l] = scale;
            s->level_res[l] = res;
            s->level_size[l] = (uint32_t)size;
            s->level_offset[l] = (uint32_t)off;
            s->level_hashed[l] = dense > size ? 1u : 0u;
            off += size;
        }
        LNR_REQUIRE(off * (uint64_t)s->n_features < (1ull << 30), "encoding table too large (the kernels address it with 32-bit byte offsets: < 4 GB)");
        s->enc_dim = s->n_levels * s->n_features;
        s->n_params = (int64_t)off * s->n_features;
    } else if (s->encoding == LNR_ENC_FREQUENCY) {
        LNR_REQUIRE(s->n_frequencies >= 1 && s->n_frequencies <= 32, "n_frequencies must be in [1,32]");
        s->enc_dim = 3 * 2 * s->n_frequencies;
        s->n_params = 0;
    } else {
        lnr_set_error("unknown encoding %d", s->encoding);
        return LNR_ERR_INVALID_ARG;
    }
    s->in_dim = ((s->enc_dim + 15) / 16) * 16;
    s->n_mlp_params = s->n_neurons * s->in_dim + (s->n_hidden - 1) * s->n_neurons * s->n_neurons + 16 * s->n_neurons;
    s->n_params += s->n_mlp_params;
    return LNR_OK;
}
