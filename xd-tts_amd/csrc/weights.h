// weights.h -- canonical Tacotron2 tensor table, synthetic init, on-disk container and the
// device-side packed layouts the kernels stream.
#pragma once
#include "common.h"

namespace xdtts {

struct TensorInfo {
  char name[64];
  int ndim;
  int dims[3];
  size_t numel, offset;
  float bound;  // synthetic init: U(-bound, bound)
  int kind;     // 0 plain, 1 bn.weight, 2 bn.bias, 3 bn.running_mean, 4 bn.running_var
  int rec;      // recurrent matrix (scaled by rec_scale in synthetic init)
};

const std::vector<TensorInfo> &tensor_table();
size_t tensor_total();
int tensor_index(const char *name);

// Canonical host blob (flat fp32, tensor_table() order).
void synthetic_blob(uint32_t seed, float rec_scale, std::vector<float> &blob);
void load_container(const std::string &dir, std::vector<float> &blob);
// onnx_load.cpp: the reference's own model directory (encoder.onnx, decoder_iter.onnx, postnet.onnx)
bool onnx_model_dir(const std::string &dir);
void load_onnx_dir(const std::string &dir, std::vector<float> &blob);
std::string describe_onnx_dir(const std::string &dir);
// Tacotron2::load(dir): `dir`/tacotron2.xdtw if present, else the three ONNX graphs
void load_model_dir(const std::string &dir, std::vector<float> &blob);
void save_container(const std::string &dir, const std::vector<float> &blob);

// One conv1d(+BN eval) layer lowered to an NT GEMM operand: W[co][k*ci] with BN folded in.
struct ConvGemm {
  DevBuf<float> w, b;
  int co = 0, ci = 0, k = 0;
};

// Device-resident, kernel-ready weights.  Everything stays fp32 (parity bar 1e-4 RMS).
#ifndef XDTTS_MFMA_WAVES
#define XDTTS_MFMA_WAVES 8  // (52 chunks: 46.9 us per iteration with 8, 47.2 with 16, 49.7 with 4)
#endif
constexpr int MFMA_WAVES = XDTTS_MFMA_WAVES;     // waves per block of the batched LSTM kernel: each takes 1/8 of K
#ifndef XDTTS_BATCH_MFMA_MIN
#define XDTTS_BATCH_MFMA_MIN 5
#endif
constexpr int BATCH_MFMA_MIN = XDTTS_BATCH_MFMA_MIN;  // chunks in lock-step from which the LSTMs run as MFMA GEMMs (measured: 38 us per iteration at 5..8 chunks against 41..49 us for the GEMV kernels)

struct DeviceWeights {
  DevBuf<float> emb;                         // [148][512]
  ConvGemm enc_conv[ENC_CONVS];              // [512][5*512]
  DevBuf<float> enc_wih[2], enc_bias[2];     // [1024][512], b_ih+b_hh [1024]
  DevBuf<float> enc_whhT[2];                 // [256][1024]  (transposed: coalesced over gate rows)
  DevBuf<float> mem_w;                       // [128][512]
  DevBuf<float> pre0T, pre1T;                // [80][256], [256][256] (transposed)
  DevBuf<float> att_w, att_b;                // [1024 units][4 gates][1792], [1024][4]
  DevBuf<float> q_w, v_w, loc_conv, loc_denseT;  // [128][1024], [128], [2][31][32] (transposed), [32][128]
  DevBuf<float> loc_fused;                   // [2*31 taps][128]  location dense . conv folded (persistent decoder)
  DevBuf<float> dec_w, dec_b;                // [1024][4][2560], [1024][4]
  DevBuf<float> proj_w, proj_b;              // [81][1536] (row 80 = gate), [81]
  DevBuf<float> ctx_w;                       // [CTXF_ROWS][512] context columns of att_w, dec_w, proj_w (persistent decoder's fold GEMM)
  // layouts for the partial-product epilogues of the LSTM kernels (decoder.hip):
  DevBuf<float> q_w4;                        // [256 blk][128 a][4]   = W_q[a][4 blk + i]
  DevBuf<float> proj_wh4;                    // [256 blk][84 m][4]    = W_p[m][4 blk + i]   (m < 81, padded to 84)
  DevBuf<float> proj_wc;                     // [8 cblk][81 m][64]    = W_p[m][1024 + 64 cblk + c]
  ConvGemm post_conv[POST_CONVS];
  // MFMA-fragment layout of the two LSTM weight matrices for the batched decoder path
  // (decoder.hip: k_lstm_mfma): [256 blk][16 waves][KW/16][64 lanes][4] with
  //   value = W[row 16 blk + (lane & 15)][wave KW + 16 jj + 4 (lane >> 4) + c],  KW = cols / 16,
  // so one wave-level 16-byte load is the A operand of four v_mfma_f32_16x16x4_f32.  Built lazily
  // (first batch of >= BATCH_MFMA_MIN chunks): a second 71 MB copy that B = 1 users never pay for.
  DevBuf<float> att_wm, dec_wm;
  void upload(const std::vector<float> &blob, hipStream_t s);
  void ensure_batched_layout(const std::vector<float> &blob, hipStream_t s);
};

}  // namespace xdtts
