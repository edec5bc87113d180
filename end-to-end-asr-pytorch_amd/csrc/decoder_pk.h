// Persistent attend-and-spell loops (decoder_pk.hip): ONE launch for all L decode steps instead of 4 launches per step.
#pragma once
#include "las_common.h"

// Bytes of las_dec_state.pk_ws a shape needs; 0 = the shape / mode is not handled by the persistent kernels (the
// per-step launch path of decoder.hip / decoder_bwd.hip runs instead).
size_t las_dec_pk_fwd_ws_bytes(const las_dec_dims* d);
// All L teacher-forced steps (dot or location-aware attention, one Speller layer, no dropout).  Preconditions, established by
// decoder_run: hs/cs slot 0 zeroed, att slot 0 = uniform attention, xin[:, :, 0:C] = embeddings of the fed tokens.
int las_dec_pk_fwd(const las_dec_dims* d, const las_dec_params* p, const float* enc, const float* psi,
                   const int32_t* enc_len, las_dec_state* st, hipStream_t stream);

// The same for the backward loop (decoder_pk_bwd.hip); fills dgates, dxin[:, :, C:] (the context half), dq_pre, de, df of
// las_dec_bwd_state; the embedding half of dxin follows from las_dec_pk_bwd_emb.
size_t las_dec_pk_bwd_ws_bytes(const las_dec_dims* d);
int las_dec_pk_bwd(const las_dec_dims* d, const las_dec_params* p, const float* enc, const float* psi, const int32_t* enc_len,
                   const las_dec_state* st, const float* g_htop, las_dec_bwd_state* bw, hipStream_t stream);
// dxin[:, :, 0:C] (the embedding half) of the persistent backward loop's result: one GEMM over the saved d gates.
int las_dec_pk_bwd_emb(const las_dec_dims* d, const las_dec_params* p, las_dec_bwd_state* bw, hipStream_t stream);
