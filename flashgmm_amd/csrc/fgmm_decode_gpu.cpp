// fgmm_decode_gpu.cpp — checkpointed bitstreams decoded ON THE GPU (segdec_kernel, fgmm_tab.hip): one workgroup per segment between
// two notes of the encoder, no decode-side tables, nothing but the bitstreams and their notes crosses PCIe.  Every segment is
// verified against the next note; a bitstream with a segment the kernel does not settle goes back to the table path
// (fgmm_decode.cpp).  What it replaces: RansDecoder::decode_with_indexes_gmm, rans_interface.cpp:766-883, for streams that carry
// out-of-band checkpoints (include/flashgmm_amd.h: fgmm_ckpt).
#include "fgmm_ctx.h"

namespace fgmm {

// can this item be decoded by the GPU's segment decoder?  (checkpoints exactly as an encoder notes them for this many symbols;
// a float latent to write; a half-width whose window fits the kernel's 16-bit fields)
bool gpu_decodable(const DecItem &it, int64_t n) {
  return it.y_hat && !it.sym_host_out && it.ckpt && it.n_ckpt > 0 && it.ckpt_stride >= 256 && !(it.ckpt_stride & (it.ckpt_stride - 1)) &&
         n > 0 && it.n_ckpt == (n - 1) / it.ckpt_stride && it.n_ckpt < (1 << 24) && it.max_bs >= 0 && 2 * (int64_t)it.max_bs + 2 <= 2048 /* kSegCapE: a latent's window in the wave's LDS */ &&
         it.enc_len >= 8 && !(it.enc_len & 3) && it.stride_p == 1;
}

// Checkpointed bitstreams decoded ON THE GPU (segdec_kernel: one workgroup of two or three waves per segment, no tables, nothing but the bitstreams and
// their notes crosses PCIe).  `which`: the items to decode; on return `redo` holds those whose segments did not all verify
// (a row the kernel leaves to the reference's bisection, wrong notes): the caller sends them through the table path.
int decode_batch_gpu(fgmm_ctx *ctx, dev::Stream stream, std::vector<DecItem> &items, const std::vector<int> &which, int mode,
                     std::vector<int> &redo) {
  Trace tr("decode-gpu", (int)ctx->opt.trace);
  const int count = (int)which.size();
  const bool clamped = items[which[0]].clamp != 0, f16 = items[which[0]].prm.dtype == FGMM_F16;
  // ---- device workspace: [descs][segment list][per item: channel list | status | checkpoints | bitstream]
  Arena ar;
  int64_t n_segs = 0;
  for (int k = 0; k < count; ++k) n_segs += items[which[k]].n_ckpt + 1;
  const size_t o_descs = ar.take(sizeof(SegDesc) * (size_t)count);
  const size_t o_segs = ar.take(sizeof(SegRef) * (size_t)n_segs);
  struct Off {
    size_t list, ckpt, words, status;
  };
  std::vector<Off> off((size_t)count);
  for (int k = 0; k < count; ++k) {
    DecItem &it = items[which[k]];
    off[(size_t)k].list = ar.take(sizeof(int32_t) * (size_t)std::max(it.M, 1), 16); // live channels, then dead ones
    off[(size_t)k].ckpt = ar.take(sizeof(fgmm_ckpt) * (size_t)it.n_ckpt, 16);
    off[(size_t)k].words = ar.take(it.enc_len, 16);
  }
  const size_t upload_bytes = ar.off;
  const size_t o_status = ar.take(sizeof(uint32_t) * (size_t)n_segs, 256);
  {
    size_t at = o_status;
    for (int k = 0; k < count; ++k) {
      off[(size_t)k].status = at;
      at += sizeof(uint32_t) * (size_t)(items[which[k]].n_ckpt + 1);
    }
  }
  int rc;
  if ((rc = ctx->ensure_device(ar.off)) || (rc = ctx->ensure_host(ar.off))) return rc;
  SegDesc *hd = reinterpret_cast<SegDesc *>(ctx->h_ws + o_descs);
  SegRef *hs = reinterpret_cast<SegRef *>(ctx->h_ws + o_segs);
  int64_t max_dead = 0;
  for (int k = 0; k < count; ++k) {
    DecItem &it = items[which[k]];
    int32_t *list = reinterpret_cast<int32_t *>(ctx->h_ws + off[(size_t)k].list);
    int r = 0, dead = it.n_ch;
    for (int c = 0; c < it.M; ++c) {
      if (!it.zero_bitmap || it.zero_bitmap[c] != 0) list[r++] = c;
      else list[dead++] = c;
    }
    max_dead = std::max<int64_t>(max_dead, it.M - it.n_ch);
    memcpy(ctx->h_ws + off[(size_t)k].ckpt, it.ckpt, sizeof(fgmm_ckpt) * (size_t)it.n_ckpt);
    memcpy(ctx->h_ws + off[(size_t)k].words, it.enc, it.enc_len);
    SegDesc &d = hd[k];
    memset(&d, 0, sizeof d);
    d.scales = it.prm.scales;
    d.means = it.prm.means;
    d.weights = it.prm.weights;
    d.stride_k = it.prm.stride_k;
    d.stride_c = it.prm.stride_c;
    d.stride_p = it.stride_p;
    d.hw = it.hw;
    d.n = it.n;
    d.chan_list = reinterpret_cast<const int32_t *>(ctx->d_ws + off[(size_t)k].list);
    d.max_bs = it.max_bs;
    d.clamp = it.clamp;
    d.logits = (it.prm.flags & FGMM_PARAMS_LOGITS) ? 1 : 0;
    d.words = reinterpret_cast<const uint32_t *>(ctx->d_ws + off[(size_t)k].words);
    d.n_words = (int64_t)(it.enc_len / 4);
    d.ckpt = reinterpret_cast<const fgmm_ckpt *>(ctx->d_ws + off[(size_t)k].ckpt);
    d.n_ckpt = it.n_ckpt;
    d.stride = it.ckpt_stride;
    d.y_hat = it.y_hat;
    d.status = reinterpret_cast<uint32_t *>(ctx->d_ws + off[(size_t)k].status);
    d.dead_list = d.chan_list + it.n_ch; // channels without a coded symbol are zero in y_hat (entropy_models.py:903-908)
    d.n_dead = it.M - it.n_ch;
  }
  // The segments in the order of the launch's workgroups: HEAVIEST FIRST.  A segment costs its symbols plus its edges, and a latent's
  // window is wide where its symbol is expensive - so the words a segment takes of the bitstream (the distance between its notes) rank
  // the segments by weight; with the heavy ones (3x the median on Kodak-like latents) in front, the launch does not end on one that
  // started last.  (The notes are not trusted: a wrong one spoils an order, nothing else.)
  {
    constexpr int kBuckets = 256;
    std::vector<uint32_t> wgt((size_t)n_segs);
    uint32_t w_max = 1;
    int64_t at = 0;
    for (int k = 0; k < count; ++k) {
      const DecItem &it = items[which[k]];
      const uint64_t end_all = it.enc_len / 4 - 2;
      uint64_t prev = 0;
      for (int64_t sgm = 0; sgm <= it.n_ckpt; ++sgm) {
        const uint64_t pos = sgm < it.n_ckpt ? it.ckpt[sgm].pos : end_all;
        const uint64_t wds = pos >= prev ? pos - prev : 0;
        wgt[(size_t)at] = (uint32_t)std::min<uint64_t>(wds, 0x7FFFFFFFu);
        w_max = std::max(w_max, wgt[(size_t)at]);
        prev = pos;
        ++at;
      }
    }
    int64_t first[kBuckets + 1] = {};
    auto bucket = [&](uint32_t wv) { return kBuckets - 1 - (int)((uint64_t)wv * (kBuckets - 1) / w_max); }; // heavy -> bucket 0
    for (int64_t q = 0; q < n_segs; ++q) ++first[bucket(wgt[(size_t)q]) + 1];
    for (int b = 0; b < kBuckets; ++b) first[b + 1] += first[b];
    at = 0;
    for (int k = 0; k < count; ++k)
      for (int64_t sgm = 0; sgm <= items[which[k]].n_ckpt; ++sgm, ++at) hs[first[bucket(wgt[(size_t)at])]++] = SegRef{k, (int32_t)sgm};
  }
  DEV_TRY(dev::copy_async(ctx->d_ws, ctx->h_ws, upload_bytes, dev::kH2D, stream));
  LAUNCH_TRY(launch_segzero(reinterpret_cast<const SegDesc *>(ctx->d_ws + o_descs), count, max_dead, stream));
  if ((rc = ctx->prof_begin(3, stream))) return rc;
  LAUNCH_TRY(launch_segdec(reinterpret_cast<const SegDesc *>(ctx->d_ws + o_descs), reinterpret_cast<const SegRef *>(ctx->d_ws + o_segs), n_segs, mode,
                           clamped, f16, stream));
  if ((rc = ctx->prof_end(3, stream))) return rc;
  DEV_TRY(dev::copy_async(ctx->h_ws + o_status, ctx->d_ws + o_status, sizeof(uint32_t) * (size_t)n_segs, dev::kD2H, stream));
  tr.mark("enqueued");
  const double t_enq = tr.ms();
  DEV_TRY(dev::stream_sync(stream));
  tr.mark("segments decoded");
  {
    const double t_done = tr.ms(), mk[5] = {t_enq, t_enq, t_enq, t_done, t_done};
    ctx->log_call(2, count, tr, mk, 0.0, 0.0);
  }
  for (int k = 0; k < count; ++k) {
    DecItem &it = items[which[k]];
    const uint32_t *st = reinterpret_cast<const uint32_t *>(ctx->h_ws + off[(size_t)k].status);
    uint32_t worst = 0;
    for (int64_t sgm = 0; sgm <= it.n_ckpt; ++sgm) worst = std::max(worst, st[sgm]);
    if (worst != kSegOk) redo.push_back(which[k]);
    else it.status = FGMM_OK, it.done.store(1);
    if (tr.on && worst != kSegOk) {
      int64_t first = 0, n_bad = 0;
      for (int64_t sgm = it.n_ckpt; sgm >= 0; --sgm)
        if (st[sgm] != kSegOk) first = sgm, ++n_bad;
      fprintf(stderr, "[fgmm decode-gpu]   item %d: %lld of %lld segments not ok, the first: segment %lld status %u\n", which[k], (long long)n_bad,
              (long long)it.n_ckpt + 1, (long long)first, st[first]);
    }
  }
  ctx->stat[1] = 0; // no decode-side tables at all
  ctx->stat[2] = ctx->stat[3] = 0;
  ctx->stat[4] = (unsigned long long)(count - (int)redo.size());
  ctx->stat[5] = (unsigned long long)redo.size();
  for (int k = 0; k < count; ++k) ctx->stat[2] += (unsigned long long)items[which[k]].n;
  return FGMM_OK;
}

} // namespace fgmm
