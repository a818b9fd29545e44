// dp_unity.cpp -- libDragPoserDLL.so: the reference's ten-function Unity plugin ABI (DragPoserDLL/exportFunc.h:61-70)
// implemented natively on top of the dp_* library.  What the reference does in python/src/run_drag.py (RunDrag) and in
// DragPose.set_initial_pose / the epilogue of DragPose.run is restated here in C++; the optimise loop itself is
// dp_optimize on the GPU.  No HIP headers in this file: device buffers go through the dp_io_* helpers.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/dragposer.h"
#include "../../include/dragposer_unity.h"

namespace {

constexpr int NJ = DP_NUM_JOINTS, LAT = DP_LATENT;

struct Tensor { std::vector<unsigned> dims; std::vector<float> data; };

struct Quat { float w, x, y, z; };
Quat qmul(Quat a, Quat b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
Quat qconj(Quat q) { return {q.w, -q.x, -q.y, -q.z}; }
void qmat(Quat q, float* m) { // pymotion quat.to_matrix == utils.to_matrix_4's 3x3 block
    m[0] = 1 - 2 * (q.y * q.y + q.z * q.z); m[1] = 2 * (q.x * q.y - q.w * q.z); m[2] = 2 * (q.x * q.z + q.w * q.y);
    m[3] = 2 * (q.x * q.y + q.w * q.z); m[4] = 1 - 2 * (q.x * q.x + q.z * q.z); m[5] = 2 * (q.y * q.z - q.w * q.x);
    m[6] = 2 * (q.x * q.z - q.w * q.y); m[7] = 2 * (q.y * q.z + q.w * q.x); m[8] = 1 - 2 * (q.x * q.x + q.y * q.y);
}

} // namespace

struct DragPoser {
    std::string err;
    // skeleton (set_reference_skeleton)
    std::vector<int> parents;
    std::vector<float> offsets;
    // model (load_models)
    std::map<std::string, Tensor> tensors;
    dp_ctx* ctx = nullptr;
    dp_temporal* temporal = nullptr; // the temporal predictor, when <modelPath>/temporal.bin exists (tools/export_temporal_bin.py)
    std::vector<float> mean_q, std_q; // [88]
    // tracker set
    std::vector<int> mask_idx;
    std::vector<float> weights; // [E][2]
    // optimiser parameters (run_drag.py:98-120)
    float stop_eps_pos = 1e-2f, stop_eps_rot = 1e-2f, lr = 1e-3f, lambda_rot = 1.f, lambda_tmp = 1.f;
    int max_iter = 100, window = 60;
    bool warned_temporal = false;
    int current_index = 0;   // frame inside the temporal window (drag_pose.py:399-402)
    int target_window = -1;  // window the device target buffer was sized for
    // state (drag_pose.py:47-64)
    bool initialised = false;
    float latent[LAT] = {0}, cur_pos[3] = {0, 0, 0};
    Quat cur_rot = {1, 0, 0, 0};
    int last_iters = 0;
    // device staging: one input block, one output block
    void* d_in = nullptr;
    void* d_out = nullptr;
    float* h_in = nullptr;   // page-locked staging for a frame's inputs / results (dp_io_alloc_host): the two copies of a frame are
    float* h_out = nullptr;  // asynchronous DMAs instead of blocking copies through the runtime's staging buffer
    // device state of the sequence (drag_pose.py:47-64): global position / rotation and the three 60-frame history
    // buffers, advanced by dp_sequence_advance; the temporal target buffer [window + 1][24]
    void* d_state = nullptr;
    void* d_target = nullptr;

    int fail(const std::string& m) { err = m; std::fprintf(stderr, "[DragPoserDLL] %s\n", m.c_str()); return -1; }
};

namespace {

// floats per frame in the staging blocks
constexpr int IN_Z0 = 0, IN_ZT = 24, IN_ROT = 48, IN_TP = 52, IN_TR = IN_TP + NJ * 3, IN_W = IN_TR + NJ * 9, IN_TRK = IN_W + NJ * 2,
              IN_FLOATS = IN_TRK + 8; // tracked: 22 bytes in 8 floats
constexpr int HIST = 60, NHGT = 6; // drag_pose.py:37-41, run_drag.py:153
constexpr int ST_POS = 0, ST_ROT = 4, ST_LAT = 8, ST_DISP = ST_LAT + HIST * LAT, ST_HGT = ST_DISP + HIST * 3, ST_POSE = ST_HGT + HIST * NHGT,
              ST_GPOS = ST_POSE + 88, ST_FLOATS = ST_GPOS + 4;
constexpr int OUT_Z = 0, OUT_ZPRE = 24, OUT_POSE = 48, OUT_DISP = 136, OUT_WD = 140, OUT_WR = 144, OUT_POS = 148, OUT_LOSS = OUT_POS + NJ * 3,
              OUT_ITERS = OUT_LOSS + 4, OUT_FLOATS = OUT_ITERS + 4;

bool parse_bvh_skeleton(const std::string& path, std::vector<int>& parents, std::vector<float>& offsets, std::string& err)
{
    std::ifstream f(path);
    if (!f) { err = "cannot open " + path; return false; }
    std::vector<int> stack;
    int pending = -1;
    bool in_end = false;
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string tok;
        if (!(ss >> tok)) continue;
        if (tok == "MOTION") break;
        if (tok == "ROOT" || tok == "JOINT") {
            pending = (int)parents.size();
            parents.push_back(stack.empty() ? 0 : stack.back());
            offsets.insert(offsets.end(), {0.f, 0.f, 0.f});
        } else if (tok == "End") {
            in_end = true; pending = -1;
        } else if (tok == "{") {
            stack.push_back(pending >= 0 ? pending : -2);
            pending = -1;
        } else if (tok == "}") {
            if (!stack.empty()) { if (stack.back() == -2) in_end = false; stack.pop_back(); }
        } else if (tok == "OFFSET" && !in_end && !stack.empty() && stack.back() >= 0) {
            float x, y, z;
            ss >> x >> y >> z;
            const int j = stack.back();
            offsets[3 * j] = x; offsets[3 * j + 1] = y; offsets[3 * j + 2] = z;
        }
    }
    if (parents.empty()) { err = "no joints in " + path; return false; }
    parents[0] = 0;                               // train.py:338
    offsets[0] = offsets[1] = offsets[2] = 0.f;  // train.py:340
    return true;
}

bool load_bin(const std::string& path, std::map<std::string, Tensor>& out, std::string& err)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path + " (export it with tools/export_model_bin.py)"; return false; }
    char magic[4];
    unsigned n = 0;
    f.read(magic, 4); f.read((char*)&n, 4);
    if (std::memcmp(magic, "DPM1", 4) != 0) { err = path + ": not a DPM1 model file"; return false; }
    for (unsigned i = 0; i < n && f; ++i) {
        unsigned len = 0, nd = 0;
        f.read((char*)&len, 4);
        std::string name(len, '\0');
        f.read(&name[0], len);
        f.read((char*)&nd, 4);
        Tensor t;
        t.dims.resize(nd);
        size_t cnt = 1;
        for (unsigned d = 0; d < nd; ++d) { f.read((char*)&t.dims[d], 4); cnt *= t.dims[d]; }
        t.data.resize(cnt);
        f.read((char*)t.data.data(), cnt * sizeof(float));
        out[name] = std::move(t);
    }
    if (!f) { err = path + ": truncated"; return false; }
    return true;
}

// encoder (autoencoder.py:56-143): 3 x [masked dense -> pool -> LeakyReLU], f_mu / f_logvar; input 176 -> mu, logvar [24]
bool encode(const std::map<std::string, Tensor>& T, const std::vector<float>& pose, float* mu, float* logvar, std::string& err)
{
    std::vector<float> h = pose;
    for (int l = 0; l < 3; ++l) {
        const std::string p = "encoder.layers." + std::to_string(l);
        auto w = T.find(p + ".0.weight"), m = T.find(p + ".0.mask"), b = T.find(p + ".0.bias"), pool = T.find(p + ".1.weight");
        if (w == T.end() || m == T.end() || b == T.end() || pool == T.end()) { err = "model file has no encoder tensors"; return false; }
        const unsigned no = w->second.dims[0], ni = w->second.dims[1];
        std::vector<float> c(no);
        for (unsigned i = 0; i < no; ++i) {
            float s = b->second.data[i];
            for (unsigned k = 0; k < ni; ++k) s += w->second.data[i * ni + k] * m->second.data[i * ni + k] * h[k];
            c[i] = s;
        }
        const unsigned po = pool->second.dims[0], pi = pool->second.dims[1];
        h.assign(po, 0.f);
        for (unsigned i = 0; i < po; ++i) {
            float s = 0;
            for (unsigned k = 0; k < pi; ++k) s += pool->second.data[i * pi + k] * c[k];
            h[i] = s > 0 ? s : 0.2f * s;
        }
    }
    const Tensor &mw = T.at("encoder.f_mu.weight"), &mb = T.at("encoder.f_mu.bias"), &lw = T.at("encoder.f_logvar.weight"),
                 &lb = T.at("encoder.f_logvar.bias");
    for (int i = 0; i < LAT; ++i) {
        float a = mb.data[i], c = lb.data[i];
        for (size_t k = 0; k < h.size(); ++k) { a += mw.data[i * h.size() + k] * h[k]; c += lw.data[i * h.size() + k] * h[k]; }
        mu[i] = a; logvar[i] = c;
    }
    return true;
}

dp_seq_state seq_state(DragPoser* d)
{
    float* st = (float*)d->d_state;
    dp_seq_state s;
    std::memset(&s, 0, sizeof(s));
    s.global_pos = st + ST_POS; s.global_rot = st + ST_ROT; s.latent_buf = st + ST_LAT; s.disp_buf = st + ST_DISP; s.heights_buf = st + ST_HGT;
    s.history = HIST; s.n_heights = NHGT;
    const int hj[NHGT] = {0, 4, 8, 13, 17, 21}; // run_drag.py:153
    for (int i = 0; i < NHGT; ++i) s.height_joints[i] = hj[i];
    return s;
}

// device state as DragPose.set_initial_pose leaves it (drag_pose.py:47-64): history filled with the initial latent,
// zero displacements, the initial heights (the Unity path passes zeros, run_drag.py:93)
bool reset_device_state(DragPoser* d)
{
    std::vector<float> st(ST_FLOATS, 0.f);
    for (int a = 0; a < 3; ++a) st[ST_POS + a] = d->cur_pos[a];
    st[ST_ROT] = d->cur_rot.w; st[ST_ROT + 1] = d->cur_rot.x; st[ST_ROT + 2] = d->cur_rot.y; st[ST_ROT + 3] = d->cur_rot.z;
    for (int t = 0; t < HIST; ++t) std::memcpy(&st[ST_LAT + t * LAT], d->latent, sizeof(d->latent));
    d->current_index = 0;
    return dp_io_upload(d->ctx, d->d_state, st.data(), st.size() * sizeof(float), nullptr) == DP_OK &&
           dp_io_upload(d->ctx, (float*)d->d_out + OUT_Z, d->latent, sizeof(d->latent), nullptr) == DP_OK && // the latent is device-resident (in/out of every frame)
           dp_stream_sync(d->ctx, nullptr) == DP_OK;
}

// <modelPath>/temporal.bin (DPM1, written by tools/export_temporal_bin.py from the reference's temporal.pt): the state_dict of
// Temporal (temporal_transformer.py) + means_latent, stds_latent, sample_step
bool load_temporal(DragPoser* d, const std::string& path, std::string& err)
{
    std::map<std::string, Tensor> T;
    if (!load_bin(path, T, err)) return false;
    auto get = [&](const std::string& k) -> const float* { auto it = T.find(k); if (it == T.end()) { err = path + ": no tensor " + k; return nullptr; } return it->second.data.data(); };
    auto has = [&](const std::string& k) { return T.find(k) != T.end(); };
    dp_temporal_model m;
    std::memset(&m, 0, sizeof(m));
    int n_enc = 0, n_dec = 0;
    while (has("temporal.encoder.layers." + std::to_string(n_enc) + ".linear1.weight")) ++n_enc;
    while (has("temporal.decoder.layers." + std::to_string(n_dec) + ".linear1.weight")) ++n_dec;
    if (n_enc < 1 || n_dec < 1 || n_enc > DP_TEMPORAL_MAX_LAYERS || n_dec > DP_TEMPORAL_MAX_LAYERS || !has("in_proj_encoder.weight") ||
        !has("positional_encoding.pos_encoding")) { err = path + ": not a Temporal state_dict"; return false; }
    const Tensor& ipe = T["in_proj_encoder.weight"];
    m.n_heights = (int)ipe.dims[1] - LAT - 3;
    m.dim_feedforward = (int)T["temporal.encoder.layers.0.linear1.weight"].dims[0];
    m.n_encoder_layers = n_enc; m.n_decoder_layers = n_dec;
    m.max_len = (int)T["positional_encoding.pos_encoding"].dims[0];
    m.sample_step = has("sample_step") ? (int)T["sample_step"].data[0] : 4; // train_temporal.py:15
    std::vector<dp_temporal_layer> enc(n_enc), dec(n_dec);
    bool ok = true;
    auto fill = [&](dp_temporal_layer& L, const std::string& p, bool is_dec) {
        std::memset(&L, 0, sizeof(L));
        ok = ok && (L.sa_in_w = get(p + "self_attn.in_proj_weight")) && (L.sa_in_b = get(p + "self_attn.in_proj_bias")) &&
             (L.sa_out_w = get(p + "self_attn.out_proj.weight")) && (L.sa_out_b = get(p + "self_attn.out_proj.bias")) &&
             (L.lin1_w = get(p + "linear1.weight")) && (L.lin1_b = get(p + "linear1.bias")) && (L.lin2_w = get(p + "linear2.weight")) &&
             (L.lin2_b = get(p + "linear2.bias")) && (L.norm1_w = get(p + "norm1.weight")) && (L.norm1_b = get(p + "norm1.bias")) &&
             (L.norm2_w = get(p + "norm2.weight")) && (L.norm2_b = get(p + "norm2.bias"));
        if (is_dec)
            ok = ok && (L.ca_in_w = get(p + "multihead_attn.in_proj_weight")) && (L.ca_in_b = get(p + "multihead_attn.in_proj_bias")) &&
                 (L.ca_out_w = get(p + "multihead_attn.out_proj.weight")) && (L.ca_out_b = get(p + "multihead_attn.out_proj.bias")) &&
                 (L.norm3_w = get(p + "norm3.weight")) && (L.norm3_b = get(p + "norm3.bias"));
    };
    for (int l = 0; l < n_enc; ++l) fill(enc[l], "temporal.encoder.layers." + std::to_string(l) + ".", false);
    for (int l = 0; l < n_dec; ++l) fill(dec[l], "temporal.decoder.layers." + std::to_string(l) + ".", true);
    ok = ok && (m.in_proj_encoder_w = get("in_proj_encoder.weight")) && (m.in_proj_encoder_b = get("in_proj_encoder.bias")) &&
         (m.in_proj_decoder_w = get("in_proj_decoder.weight")) && (m.in_proj_decoder_b = get("in_proj_decoder.bias")) &&
         (m.out_proj_w = get("out_proj.weight")) && (m.out_proj_b = get("out_proj.bias")) && (m.pos_encoding = get("positional_encoding.pos_encoding")) &&
         (m.enc_norm_w = get("temporal.encoder.norm.weight")) && (m.enc_norm_b = get("temporal.encoder.norm.bias")) &&
         (m.dec_norm_w = get("temporal.decoder.norm.weight")) && (m.dec_norm_b = get("temporal.decoder.norm.bias")) &&
         (m.means_latent = get("means_latent")) && (m.stds_latent = get("stds_latent"));
    if (!ok) return false;
    if (m.n_heights != NHGT) { err = path + ": the predictor takes " + std::to_string(m.n_heights) + " heights per token, the plugin feeds 6"; return false; }
    m.enc = enc.data(); m.dec = dec.data();
    if (dp_temporal_create(&d->temporal, &m, 0) != DP_OK) { err = std::string("dp_temporal_create: ") + dp_temporal_last_error(nullptr); return false; }
    return true;
}

} // namespace

extern "C" {

DragPoser* init_drag_poser(void) { return new DragPoser(); }

const char* drag_poser_last_error(const DragPoser* d) { return d ? d->err.c_str() : "null handle"; }
int drag_poser_last_iterations(const DragPoser* d) { return d ? d->last_iters : 0; }
void drag_poser_get_latent(const DragPoser* d, float* z) { if (d && z) std::memcpy(z, d->latent, sizeof(d->latent)); }
void drag_poser_set_latent(DragPoser* d, const float* z)
{ // replaces the latent and re-initialises the history with it, as set_initial_pose does with the encoder's (drag_pose.py:50-55)
    if (!d || !z) return;
    std::memcpy(d->latent, z, sizeof(d->latent));
    if (d->ctx && d->d_state && !reset_device_state(d)) d->fail(std::string("drag_poser_set_latent: ") + dp_last_error(d->ctx));
}
int drag_poser_has_temporal(const DragPoser* d) { return d && d->temporal ? 1 : 0; }

void set_reference_skeleton(DragPoser* d, char* bvhPath)
{ // run_drag.py:30-38
    if (!d || !bvhPath) return;
    d->err.clear();
    d->parents.clear(); d->offsets.clear();
    std::string e;
    if (!parse_bvh_skeleton(bvhPath, d->parents, d->offsets, e)) { d->fail(e); return; }
    if ((int)d->parents.size() != NJ) d->fail("skeleton has " + std::to_string(d->parents.size()) + " joints, this build supports 22");
}

void load_models(DragPoser* d, char* modelPath)
{ // run_drag.py:40-60
    if (!d || !modelPath) return;
    d->err.clear();
    if ((int)d->parents.size() != NJ) { d->fail("load_models: call set_reference_skeleton first"); return; }
    std::string e;
    d->tensors.clear();
    if (!load_bin(std::string(modelPath) + "/dragposer_model.bin", d->tensors, e)) { d->fail(e); return; }
    auto& T = d->tensors;
    auto need = [&](const char* k) -> const float* { auto it = T.find(k); return it == T.end() ? nullptr : it->second.data.data(); };
    dp_model m;
    std::memset(&m, 0, sizeof(m));
    m.f_latent_w = need("decoder.f_latent.weight"); m.f_latent_b = need("decoder.f_latent.bias");
    for (int l = 0; l < 3; ++l) {
        const std::string p = "decoder.layers." + std::to_string(l);
        m.unpool_w[l] = need((p + ".0.weight").c_str()); m.conv_w[l] = need((p + ".1.weight").c_str());
        m.conv_mask[l] = need((p + ".1.mask").c_str()); m.conv_b[l] = need((p + ".1.bias").c_str());
    }
    const float *mean = need("means.dqs"), *sd = need("stds.dqs");
    if (!mean || !sd) { d->fail("model file lacks means.dqs / stds.dqs"); return; }
    d->mean_q.resize(88); d->std_q.resize(88);
    for (int j = 0; j < NJ; ++j)
        for (int c = 0; c < 4; ++c) { d->mean_q[4 * j + c] = mean[8 * j + c]; d->std_q[4 * j + c] = sd[8 * j + c]; } // drag_pose.py:27-33
    m.mean_q = d->mean_q.data(); m.std_q = d->std_q.data();
    m.mean_disp = need("means.displacement"); m.std_disp = need("stds.displacement");
    m.parents = d->parents.data(); m.offsets = d->offsets.data();
    m.weight_dtype = DP_WEIGHTS_FP32;
    if (d->ctx) { // loaded before: release the previous context and its device buffers
        for (void** p : {&d->d_in, &d->d_out, &d->d_state, &d->d_target}) { if (*p) dp_io_free(d->ctx, *p); *p = nullptr; }
        for (float** p : {&d->h_in, &d->h_out}) { if (*p) dp_io_free_host(d->ctx, *p); *p = nullptr; }
        if (d->temporal) dp_temporal_destroy(d->temporal);
        d->temporal = nullptr;
        d->target_window = -1;
        dp_destroy(d->ctx);
        d->ctx = nullptr;
        // the device state (latent / displacement / height histories, global pose) is gone with the buffers: the caller
        // has to call init_drag_model again before the next drag_pose, and the temporal window restarts
        d->initialised = false;
        d->current_index = 0;
    }
    if (dp_create(&d->ctx, &m, 0) != DP_OK) { d->fail(std::string("dp_create: ") + dp_last_error(nullptr)); return; }
    if (dp_io_alloc_host(d->ctx, IN_FLOATS * sizeof(float), (void**)&d->h_in) != DP_OK || dp_io_alloc_host(d->ctx, OUT_FLOATS * sizeof(float), (void**)&d->h_out) != DP_OK ||
        dp_io_alloc(d->ctx, IN_FLOATS * sizeof(float), &d->d_in) != DP_OK || dp_io_alloc(d->ctx, OUT_FLOATS * sizeof(float), &d->d_out) != DP_OK ||
        dp_io_alloc(d->ctx, ST_FLOATS * sizeof(float), &d->d_state) != DP_OK) {
        d->fail(std::string("device buffers: ") + dp_last_error(d->ctx));
        return;
    }
    // the temporal predictor is optional: without temporal.bin the plugin runs with the pull term off (and says so)
    const std::string tpath = std::string(modelPath) + "/temporal.bin";
    if (std::ifstream(tpath, std::ios::binary).good() && !load_temporal(d, tpath, e)) d->fail(e);
}

void set_mask_and_weights(DragPoser* d, float* mask, dp_float2* weights)
{ // run_drag.py:62-77
    if (!d || !mask || !weights) return;
    d->err.clear();
    d->mask_idx.clear(); d->weights.clear();
    for (int j = 0; j < (int)d->parents.size(); ++j)
        if (mask[j] != 0.f) { d->mask_idx.push_back(j); d->weights.push_back(weights[j].x); d->weights.push_back(weights[j].y); }
}

void init_drag_model(DragPoser* d, dp_float3 pos, dp_quaternion rot)
{ // run_drag.py:79-96 + drag_pose.py:47-64: the encoder sees an all-zero normalised pose
    if (!d) return;
    d->err.clear();
    if (!d->ctx) { d->fail("init_drag_model: call load_models first"); return; }
    float mu[LAT], logvar[LAT];
    std::string e;
    if (!encode(d->tensors, std::vector<float>(NJ * 8, 0.f), mu, logvar, e)) { d->fail(e); return; }
    std::mt19937 gen(2222); // train.param["seed"]
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int i = 0; i < LAT; ++i) d->latent[i] = mu[i] + nd(gen) * std::exp(0.5f * logvar[i]); // autoencoder.py:19-27
    d->cur_pos[0] = pos.x; d->cur_pos[1] = pos.y; d->cur_pos[2] = pos.z;
    d->cur_rot = {rot.w, rot.x, rot.y, rot.z};
    if (!reset_device_state(d)) { d->fail(std::string("init_drag_model: ") + dp_last_error(d->ctx)); return; }
    d->initialised = true;
}

void set_optim_params(DragPoser* d, float stopEpsPos, float stopEpsRot, int maxIter, float lr)
{
    if (!d) return;
    d->stop_eps_pos = stopEpsPos; d->stop_eps_rot = stopEpsRot; d->max_iter = maxIter; d->lr = lr;
}

void set_lambdas(DragPoser* d, float lambdaRot, float lambdaTemporal, int temporalFutureWindow)
{
    if (!d) return;
    d->lambda_rot = lambdaRot; d->lambda_tmp = lambdaTemporal; d->window = temporalFutureWindow;
    d->err.clear();
    if (lambdaTemporal != 0.f && !d->temporal) {
        // The reference runs its temporal Transformer here (drag_pose.py:234-294).  Without <modelPath>/temporal.bin this
        // plugin has none: the call is accepted, the pull term stays off, and the difference is REPORTED through
        // drag_poser_last_error (the reference ABI has no return codes) -- and once on stderr.
        d->err = "set_lambdas: no temporal predictor loaded (no temporal.bin in the model folder); lambdaTemporal = " + std::to_string(lambdaTemporal) +
                 " is treated as 0 (results differ from the reference's for a non-zero lambda)";
        if (!d->warned_temporal) {
            d->warned_temporal = true;
            std::fprintf(stderr, "[DragPoserDLL] %s\n", d->err.c_str());
        }
    }
}

void set_global_pos(DragPoser* d, dp_float3 p)
{
    if (!d) return;
    d->cur_pos[0] = p.x; d->cur_pos[1] = p.y; d->cur_pos[2] = p.z;
    if (d->ctx && d->d_state && (dp_io_upload(d->ctx, (float*)d->d_state + ST_POS, d->cur_pos, sizeof(d->cur_pos), nullptr) != DP_OK ||
                                 dp_stream_sync(d->ctx, nullptr) != DP_OK))
        d->fail(std::string("set_global_pos: ") + dp_last_error(d->ctx));
}

void drag_pose(DragPoser* d, int nEE, dp_float3* tp, dp_quaternion* tq, dp_quaternion* resultPose, dp_float3* resultGlobalPos)
{ // run_drag.py:126-176 around DragPose.run with joint_adjustment_indices=None
    if (!d || !tp || !tq || !resultPose || !resultGlobalPos) return;
    d->err.clear();
    if (!d->ctx || !d->initialised) { d->fail("drag_pose: call load_models and init_drag_model first"); return; }
    if (nEE != (int)d->mask_idx.size()) { d->fail("drag_pose: nEndEffectors differs from the tracker mask"); return; }
    if (d->max_iter < 1 || d->max_iter > DP_MAX_ITERS) { // (the reference has no cap; DP_MAX_ITERS is a sanity bound)
        d->fail("drag_pose: maxIter must be in [1, " + std::to_string(DP_MAX_ITERS) + "]");
        return;
    }
    float* in = d->h_in;
    std::memset(in, 0, IN_FLOATS * sizeof(float));
    std::memcpy(in + IN_Z0, d->latent, sizeof(d->latent)); // (the staged z_tgt stays 0: used when the pull term is off)
    in[IN_ROT] = d->cur_rot.w; in[IN_ROT + 1] = d->cur_rot.x; in[IN_ROT + 2] = d->cur_rot.y; in[IN_ROT + 3] = d->cur_rot.z;
    unsigned char* trk = (unsigned char*)(in + IN_TRK);
    for (int e = 0; e < nEE; ++e) {
        const int j = d->mask_idx[e];
        in[IN_TP + 3 * j] = tp[e].x; in[IN_TP + 3 * j + 1] = tp[e].y; in[IN_TP + 3 * j + 2] = tp[e].z;
        qmat({tq[e].w, tq[e].x, tq[e].y, tq[e].z}, in + IN_TR + 9 * j);
        in[IN_W + 2 * j] = d->weights[2 * e]; in[IN_W + 2 * j + 1] = d->weights[2 * e + 1];
        trk[j] = 1;
    }
    float* di = (float*)d->d_in;
    float* dout = (float*)d->d_out;
    dp_batch b;
    b.n_frames = 1;
    b.z0 = di + IN_Z0; b.z_tgt = di + IN_ZT; b.cur_rot = di + IN_ROT; b.tgt_pos = di + IN_TP; b.tgt_rot = di + IN_TR; b.w = di + IN_W;
    b.tracked = (const unsigned char*)(di + IN_TRK);
    // temporal target block (drag_pose.py:234-294): a prediction every `window` frames, one row of it per frame.
    // The reference's schedule: the buffer is (re)allocated when the window changes (drag_pose.py:238-246) and the predictor runs
    // whenever current_index == 0 -- WHATEVER lambda_temporal is (drag_pose.py:247-291): with the pull term off the buffer is still
    // refreshed every `window` frames, so a caller that switches the term on in the middle of a window (Unity's SetLambdas) pulls
    // towards the prediction made at that window's start.  This plug-in does the same (round 3 predicted only while the term was on
    // and restarted the window at switch-on: a different z_tgt for up to window - 1 frames; tests/golden/sequ_switch.npz pins the
    // reference's behaviour).  Where it still departs: a window CHANGED mid-window restarts the window (below).
    const bool have_predictor = d->temporal != nullptr;
    const bool pull = have_predictor && d->lambda_tmp != 0.f;
    dp_seq_state st = seq_state(d);
    // (the sequence's global position / rotation live on the device: reset_device_state, set_global_pos; the staged
    //  cur_rot below is what dp_optimize took and is kept for reference)
    if (have_predictor) {
        if (d->window < 0) { d->fail("drag_pose: temporalFutureWindow must not be negative"); return; } // (a multiple of the predictor's sample_step: dp_temporal_predict checks)
        if (d->target_window != d->window) { // (re)sized and zeroed, as drag_pose.py:238-246 does
            // Unity's SetLambdas may change the window mid-window.  The reference keeps current_index and then indexes the
            // re-allocated (window + 1)-row buffer with it: an IndexError when the window shrank, rows of zeros until the
            // next prediction when it grew.  Here a changed window restarts the window: a prediction is made this frame.
            d->current_index = 0;
            if (d->d_target) dp_io_free(d->ctx, d->d_target);
            d->d_target = nullptr;
            const std::vector<float> zeros((size_t)(d->window + 1) * LAT, 0.f);
            if (dp_io_alloc(d->ctx, zeros.size() * sizeof(float), &d->d_target) != DP_OK ||
                dp_io_upload(d->ctx, d->d_target, zeros.data(), zeros.size() * sizeof(float), nullptr) != DP_OK ||
                dp_stream_sync(d->ctx, nullptr) != DP_OK) { d->fail(std::string("drag_pose: ") + dp_last_error(d->ctx)); return; }
            d->target_window = d->window;
        }
        if (d->current_index == 0 && dp_temporal_predict(d->temporal, 1, &st, d->window, (float*)d->d_target, nullptr) != DP_OK) {
            d->fail(std::string("drag_pose: ") + dp_temporal_last_error(d->temporal));
            return;
        }
        if (d->current_index > d->window) { d->fail("drag_pose: temporal window index out of range"); return; } // (cannot happen: see above)
        b.z_tgt = (float*)d->d_target + (size_t)d->current_index * LAT;
    }
    dp_params p = DP_PARAMS_INIT;
    p.n_iter = d->max_iter; p.lr = d->lr; p.beta1 = 0.9f; p.beta2 = 0.999f; p.eps = 1e-8f;
    p.lambda_rot = d->lambda_rot; p.lambda_tmp = pull ? d->lambda_tmp : 0.f;
    p.early_stop = 1; p.stop_eps_pos = d->stop_eps_pos; p.stop_eps_rot = d->stop_eps_rot; p.min_loss_incr = 0.00001f; // run() default
    p.max_trackers = 0;
    p.kernel = DP_KERNEL_AUTO;
    // one frame = a whole-sequence launch of one step: the optimise loop with the reference's while-condition and run()'s epilogue
    // (drag_pose.py:296-402) in one kernel -- the path dragposer_amd.DragPose.run takes, so the two agree bit for bit --, then the
    // history buffers (no joint adjustment on this path: run_drag.py:154)
    dp_seq_frames fr;
    std::memset(&fr, 0, sizeof(fr));
    fr.n_steps = 1;
    fr.tgt_pos = b.tgt_pos; fr.tgt_rot = b.tgt_rot; fr.tgt_root = nullptr; fr.w = b.w; fr.tracked = b.tracked;
    fr.z_tgt = b.z_tgt; fr.z_tgt_step = 0; fr.z_tgt_seq = LAT;
    // everything the host needs back lands in ONE device block (one download per frame): the latent (device-resident, in/out),
    // the returned pose, the global position and rotation after the step, the iteration count
    dp_seq_results r = DP_SEQ_RESULTS_INIT;
    r.pose_ret = dout + OUT_POSE; r.pos_ret = dout + OUT_WD; r.world_rot = dout + OUT_WR; r.iters = (int*)(dout + OUT_ITERS); r.loss = dout + OUT_LOSS;
    r.hist_scratch = dout + OUT_POS; // (LAT + 3 + NHGT floats of scratch: the block's joint-position area is unused on this path)
    static_assert(NJ * 3 >= LAT + 3 + NHGT, "the history scratch row fits");
    dp_seq_step step;
    std::memset(&step, 0, sizeof(step));
    step.adjust_joint = -1; step.adjust_target_joint = -1;
    float* out = d->h_out;
    if (dp_io_upload(d->ctx, di, in, IN_FLOATS * sizeof(float), nullptr) != DP_OK ||
        dp_optimize_sequence(d->ctx, 1, dout + OUT_Z, &fr, &p, &st, &step, &r, nullptr) != DP_OK ||
        dp_io_download(d->ctx, out, dout, OUT_FLOATS * sizeof(float), nullptr) != DP_OK || dp_stream_sync(d->ctx, nullptr) != DP_OK) {
        const std::string why = dp_last_error(d->ctx);
        dp_stream_sync(d->ctx, nullptr); // h_in / h_out are page-locked: an upload or download queued before the failure is an asynchronous DMA, and
                                         // the next drag_pose() rewrites h_in -- nothing may still be in flight when this one returns
        d->fail("drag_pose: " + why);
        return;
    }
    // the state the kernel left (drag_pose.py:369-371), mirrored on the host
    std::memcpy(d->latent, out + OUT_Z, sizeof(d->latent));
    for (int a = 0; a < 3; ++a) d->cur_pos[a] = out[OUT_WD + a];
    d->cur_rot = {out[OUT_WR], out[OUT_WR + 1], out[OUT_WR + 2], out[OUT_WR + 3]};
    std::memcpy(&d->last_iters, out + OUT_ITERS, sizeof(int));
    const float* ret = out + OUT_POSE;
    d->current_index = d->window <= 0 ? 0 : (d->current_index + 1) % d->window; // drag_pose.py:399-402
    // result (run_drag.py:161-176): de-normalised root-space quaternions with the world root -> parent-local rotations
    Quat q[NJ];
    for (int j = 0; j < NJ; ++j) {
        const float* pz = ret + 4 * j; // the returned pose (d_state + ST_POSE)
        q[j] = {pz[0] * d->std_q[4 * j] + d->mean_q[4 * j], pz[1] * d->std_q[4 * j + 1] + d->mean_q[4 * j + 1],
                pz[2] * d->std_q[4 * j + 2] + d->mean_q[4 * j + 2], pz[3] * d->std_q[4 * j + 3] + d->mean_q[4 * j + 3]};
    }
    q[0] = d->cur_rot; // drag_pose.py:394-396 (normalise / de-normalise round trip)
    for (int j = NJ - 1; j >= 1; --j) { // train.from_root_quat (train.py:409-434)
        const int par = d->parents[j];
        if (par != 0) q[j] = qmul(qconj(q[par]), q[j]);
    }
    for (int j = 0; j < NJ; ++j) resultPose[j] = {q[j].w, q[j].x, q[j].y, q[j].z};
    resultGlobalPos[0] = {d->cur_pos[0], d->cur_pos[1], d->cur_pos[2]};
}

void destroy_drag_poser(DragPoser* d)
{
    if (!d) return;
    if (d->ctx) {
        for (void* p : {d->d_in, d->d_out, d->d_state, d->d_target}) if (p) dp_io_free(d->ctx, p);
        for (float* p : {d->h_in, d->h_out}) if (p) dp_io_free_host(d->ctx, p);
        if (d->temporal) dp_temporal_destroy(d->temporal);
        dp_destroy(d->ctx);
    }
    delete d;
}

} // extern "C"
