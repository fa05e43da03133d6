// A host-memory stand-in for the HIP runtime and for the kernel launch wrappers of kernels.h (TEST INFRASTRUCTURE: linked
// only by tests/test_compile_asan.py, never by the product).  With it the WHOLE host engine -- the C ABI, the project
// front-end, the event compiler, the descriptor builder, the submission code of csrc/engine.cpp -- builds with
// g++ -fsanitize=address,undefined and runs without a GPU: "device" memory is malloc'd host memory (so every byte the host
// writes into an arena, a table buffer or a state slot is bounds-checked), copies are memcpy, streams and events are tokens,
// and a "launch" walks its descriptors the way the kernel would -- every descriptor of the table, both ends of every array a
// descriptor points to -- so that a descriptor table that is too short, a scratch array that is too small or a pointer that
// was never patched is an AddressSanitizer report instead of a silent wrong render.  Nothing is computed: outputs are zeros.
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "kernels.h"

static volatile unsigned char g_sink;
static void touch(const void* p, size_t bytes) {   // first and last byte of [p, p + bytes)
    if (!p || !bytes) return;
    const volatile unsigned char* b = (const volatile unsigned char*)p;
    g_sink ^= b[0];
    g_sink ^= b[bytes - 1];
}
static void touch_w(void* p, size_t bytes) {
    if (!p || !bytes) return;
    volatile unsigned char* b = (volatile unsigned char*)p;
    b[0] = b[0];
    b[bytes - 1] = b[bytes - 1];
}
template <class T>
static void touch_descs(const T* d, int n) { touch(d, (size_t)std::max(n, 0) * sizeof(T)); }

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned int) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t n, hipStream_t) {
    for (size_t i = 0; i < n; ++i) ((int*)d)[i] = v;
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "mock HIP error"; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
}

namespace tdk {

static void touch_terms(const InTerm* ins, uint32_t k, uint32_t frames) {
    touch(ins, (size_t)k * sizeof(InTerm));
    for (uint32_t i = 0; i < k; ++i) {
        const InTerm& t = ins[i];
        if (t.kind == 0u || t.kind == 4u) touch(t.p, (size_t)frames * sizeof(float2));
        else if (t.kind == 5u) {
            const AdsrVDesc* a = reinterpret_cast<const AdsrVDesc*>((uintptr_t)t.len);
            touch(a, sizeof(AdsrVDesc));
            touch(a->env, (size_t)frames * sizeof(float));
            touch_terms(a->ins, a->k, frames);
        } else if (t.kind == 3u) touch(t.p, ((size_t)t.len + 15) * 4);
        else touch(t.p, ((size_t)t.len + 15) * sizeof(float2));
    }
}
static void touch_interval_tab(const IntervalTab& t, uint32_t frames) {
    touch(t.istart, (size_t)t.n_int * 4);
    touch(t.ivoff, ((size_t)t.n_int + 1) * 4);
    const size_t nt = (frames + kTileFrames - 1) / kTileFrames;
    touch(t.tile_first, nt * 4);
    if (t.tile_order) {   // (costliest tiles first, compile.cpp put_intervals) a permutation of the chunk's tiles
        touch(t.tile_order, nt * 4);
        std::vector<char> seen(nt, 0);
        for (size_t i = 0; i < nt; ++i) {
            if (t.tile_order[i] >= nt || seen[t.tile_order[i]]) abort();
            seen[t.tile_order[i]] = 1;
        }
    }
}
static void touch_sum(const SumDesc* d, int n, uint32_t frames, uint32_t bl) {
    touch_descs(d, n);
    const uint32_t nb = bl ? (frames + bl - 1) / bl : 0;
    for (int i = 0; i < n; ++i) {
        touch_terms(d[i].ins, d[i].k, frames);
        touch_w(d[i].out, (size_t)frames * sizeof(float2));
        if (d[i].mode == 1u || d[i].mode >= 3u) { touch_w(d[i].peaks, (size_t)nb * 4); touch_w(d[i].init_copy, 8); touch(d[i].state, sizeof(NormState)); }
        if (d[i].mode == 2u) touch_w(d[i].peaks, (size_t)((frames + 255) / 256) * 4);
        if (d[i].mode >= 4u) { touch_w(d[i].sync, (size_t)std::max<uint32_t>(nb, 1) * 8); touch_w(d[i].host_flag, 4); }
        if (d[i].qmode) touch_w(d[i].pcm, (size_t)frames * 2 * (d[i].qmode == 1u ? 2 : 4));
        if (d[i].out_q4) touch_w(d[i].out_q4, (size_t)((frames + 3) & ~3u) * sizeof(float2));   // (the planar copy covers the last, partial four frames)
        if (d[i].rp) { touch(d[i].rp, sizeof(BandRespParam)); touch_w(d[i].rp->resp, (size_t)((frames + 255) / 256) * 4 * sizeof(double)); }
    }
}

void adsr_fill_run_consts(AdsrVDesc* d) { for (double& r : d->rcp) r = 1.0; d->tame = 0u; }
void launch_adsr_env(const AdsrVDesc* d, int n, uint32_t frames, hipStream_t) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) {
        touch_interval_tab(d[i].tab, frames);
        touch_w(d[i].env, ((size_t)frames + 1) * 4);
        if (d[i].env_tile) touch_w(d[i].env_tile, (size_t)((frames + 511) / 512) * 4);
    }
}
void launch_sine_probe(const ProbeDesc* d, int n, uint32_t frames, uint32_t n_groups, hipStream_t) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) {
        const ProbeDesc& p = d[i];
        const IntervalTab& t = p.kind ? p.syn.tab : p.sine.tab;
        touch_interval_tab(t, frames);
        touch(t.voices, (size_t)t.ivoff[t.n_int] * sizeof(float4));   // (the RAW table of a Synth vertex: one float4 per voice)
        touch(p.kind ? p.syn.out : p.sine.out, (size_t)frames * sizeof(float2));
        if (p.n_groups != n_groups || ((size_t)p.n_groups << (4 + p.stride_log2)) < frames) abort();
        touch_w(p.noise, (size_t)((frames + (1u << p.stride_log2) - 1) >> p.stride_log2) * 4);
        {   // the host-made voice ranges: every sample's inside the raw table
            const uint32_t ns = (frames + (1u << p.stride_log2) - 1) >> p.stride_log2;
            touch(p.ranges, (size_t)ns * 8);
            for (uint32_t q = 0; q < ns; ++q)
                if (p.ranges[2 * q] > p.ranges[2 * q + 1] || p.ranges[2 * q + 1] > t.ivoff[t.n_int]) abort();
        }
    }
}
void launch_sinf(const float* in, float* out, uint32_t n, int, hipStream_t) { touch(in, (size_t)n * 4); touch_w(out, (size_t)n * 4); }
void launch_debug_verify(const uint32_t* p, uint32_t n_words, const uint32_t* seg, uint32_t* report, hipStream_t) { touch(p, (size_t)n_words * 4); touch(seg, (size_t)((n_words + 63) / 64) * 4); touch_w(report, 256); }
void launch_band_audit(const AuditHead* h, int n, hipStream_t) {
    touch_descs(h, n);
    for (int i = 0; i < n; ++i) {
        touch(h[i].descs, (size_t)h[i].n * sizeof(AuditDesc));
        touch_w(h[i].host_word, 8);
        for (uint32_t j = 0; j < h[i].n; ++j) {
            const AuditDesc& a = h[i].descs[j];
            touch(a.noise, (size_t)a.n_wt * 4);
            if (a.peaks) { touch(a.peaks, (size_t)a.nb * 4); touch(a.init_copy, 4); }
        }
    }
}
static void touch_scan(const BandScanDesc* d, int n, uint32_t frames, bool chain, bool guarded) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) {
        const BandScanDesc& x = d[i];
        touch_terms(x.ins, x.k, frames);
        touch(x.stages, (size_t)x.n_stages * sizeof(BandStageDesc));
        touch_w(x.ticket, 8);
        touch_w(x.poison, (size_t)x.n_tiles * 8);
        if (chain) touch_w(x.rpoison, (size_t)x.n_tiles * 8);
        if (x.out) touch_w(x.out, (size_t)frames * sizeof(float2));
        for (uint32_t s = 0; s < x.n_stages; ++s) {
            const BandStageDesc& q = x.stages[s];
            touch_w(q.state, sizeof(BandState));
            touch_w(q.sync, (size_t)x.n_tiles * 128);
            touch(q.pw, 128 * sizeof(double));
            if (chain) touch(q.pk, 2 * kScanMaxK * sizeof(double));
            for (uint32_t p = 0; p < q.n_post && p < 3u; ++p)
                if (q.post[p].env) touch(q.post[p].env, (size_t)frames * 4);
            if (q.envt) touch(q.envt, (size_t)((frames + 511) / 512) * 4);
        }
        if (x.norm) {
            touch(x.norm, sizeof(SumDesc));
            touch_w(x.norm->peaks, (size_t)((frames + kTileFrames - 1) / kTileFrames) * 4);
            touch_w(x.norm->init_copy, 8);
            touch_w(x.norm->sync, (size_t)x.n_tiles * 8);
            touch(x.norm->state, sizeof(NormState));
            if (x.norm->out) touch_w(x.norm->out, (size_t)frames * sizeof(float2));
            if (x.norm->qmode) touch_w(x.norm->pcm, (size_t)frames * 2 * (x.norm->qmode == 1u ? 2 : 4));
        }
        if (x.noise) {   // (per wave-tile in the chain launch, per workgroup tile in k_band_scan)
            touch_w(x.noise, (size_t)(chain ? (frames + kTileFrames - 1) / kTileFrames : x.n_tiles) * 4);
            if (x.nz_sync) { touch_w(x.nz_sync, (size_t)x.n_tiles * 8); touch_w(x.nz_host, 8); }
            if (x.nz_probe) {   // (the launch's tiles evaluate the probe's samples themselves: one probed vertex, a sample per 256 frames)
                touch(x.nz_probe, sizeof(ProbeDesc));
                if (x.nz_probe->stride_log2 != 8u || x.nz_extra[0]) abort();
                const IntervalTab& t = x.nz_probe->kind ? x.nz_probe->syn.tab : x.nz_probe->sine.tab;
                touch_interval_tab(t, frames);
                touch(x.nz_probe->ranges, (size_t)((frames + 255) / 256) * 8);
            }
            for (int q = 0; q < 2; ++q)   // (sine_mode 2: the probed sine vertices' energies, nz_xcnt samples per wave-tile)
                if (x.nz_extra[q]) {
                    if (!x.nz_xcnt || x.nz_xcnt > 64u || (kTileFrames % x.nz_xcnt)) abort();
                    touch(x.nz_extra[q], (size_t)((frames + kTileFrames / x.nz_xcnt - 1) / (kTileFrames / x.nz_xcnt)) * 4);
                }
        }
    }
}
void launch_band_scan(const BandScanDesc* d, int n, uint32_t frames, uint32_t, int, hipStream_t) { touch_scan(d, n, frames, false, false); }
int band_scan_resident_capacity(int) { return 768; }
void launch_band_chain(const BandScanDesc* d, int n, uint32_t frames, uint32_t, bool guarded, hipStream_t) { touch_scan(d, n, frames, true, guarded); }
void launch_resample(const ResampleDesc& d, hipStream_t) { touch(d.in, (size_t)d.len * sizeof(float2)); touch_w(d.out, (size_t)d.nout * sizeof(float2)); }
void launch_pcm_decode(const uint8_t*, float* linear, uint32_t n, uint32_t, hipStream_t) { touch_w(linear, (size_t)n * 4); }
void launch_sample_split(const float* lin, uint32_t ch, uint32_t, uint32_t, float* l, float* r, uint32_t nl, uint32_t nr, hipStream_t) {
    touch(lin, (size_t)std::max(nl, nr) * ch * 4); touch_w(l, (size_t)nl * 4); touch_w(r, (size_t)nr * 4);
}
// (the load pipeline reads these scalars back: a peak of 1 keeps 1 / max finite)
void launch_absmax(const float* v, uint32_t n, float* out, hipStream_t) { touch(v, (size_t)n * 4); *out = 1.0f; }
void launch_abs_sum_serial(const float* v, uint32_t n, float* out, hipStream_t) { touch(v, (size_t)n * 4); *out = 1.0f; }
void launch_add_planar(const float* a, const float* b, float* o, uint32_t n, hipStream_t) { touch(a, (size_t)n * 4); touch(b, (size_t)n * 4); touch_w(o, (size_t)n * 4); }
void launch_sample_pack(const float* l, const float* r, const float* ml, const float* mr, float2* f, uint32_t n, hipStream_t) {
    touch(l, (size_t)n * 4); touch(r, (size_t)n * 4); touch(ml, 4); touch(mr, 4); touch_w(f, ((size_t)n + 15) * sizeof(float2));
}
void launch_sample_pack16(const float* l, const float* r, uint32_t* p, uint32_t n, uint32_t* flag, hipStream_t) {
    touch(l, (size_t)n * 4); touch(r, (size_t)n * 4); touch_w(p, ((size_t)n + 15) * 4); touch_w(flag, 4);
}
void launch_peak_table(const float* const* src, float* table, uint32_t n_total, uint32_t n_own, uint32_t, uint32_t, hipStream_t) {
    touch(src, (size_t)n_own * sizeof(float*)); touch_w(table, (size_t)n_total * 4);
}
static void touch_spec(const BandSpecDesc* d, int n, uint32_t frames) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) {
        const BandSpecDesc& x = d[i];
        if (x.x) abort();   /* (round 6: no interleaved copy) */ touch(x.xq4, (size_t)((frames + 3) & ~3u) * sizeof(float2)); touch_w(x.out, (size_t)frames * sizeof(float2));
        touch_w(x.state, sizeof(BandState));
        touch_w(x.seg_start, (size_t)x.nseg * 16); touch_w(x.seg_final, (size_t)x.nseg * 16); touch_w(x.seg_flags, (size_t)x.nseg * 4);
        touch_w(x.seg_x0, (size_t)x.nseg * 8); touch_w(x.jobs, (size_t)x.nseg * sizeof(BandJob)); touch_w(x.seg_job, (size_t)x.nseg * 4);
        touch_w(x.stats, 136);
        touch(x.blk_peaks, (size_t)((frames + 255) / 256) * 4);
        if (x.resp) touch(x.resp, (size_t)((frames + 255) / 256) * 4 * sizeof(double));
    }
}
void launch_band_spec(const BandSpecDesc* d, int n, uint32_t frames, uint32_t, hipStream_t) { touch_spec(d, n, frames); }
void launch_band_fix(const BandSpecDesc* d, int n, uint32_t frames, uint32_t, hipStream_t) { touch_spec(d, n, frames); }
int norm1_tiles_per_workgroup(uint32_t, uint32_t frames) {
    const uint32_t tiles = (frames + kTileFrames - 1) / kTileFrames;
    return tiles <= 1024 ? 1 : (tiles <= 2048 ? 2 : (tiles <= 4096 ? 4 : 0));
}
void launch_norm1(const SumDesc* d, int n, uint32_t frames, uint32_t, int, uint32_t, hipStream_t) { touch_sum(d, n, frames, kTileFrames); }
int sum16w_resident_capacity(int, bool) { return 1024; }
void launch_sum(const SumDesc* d, int n, uint32_t frames, uint32_t bl, uint32_t, bool, bool, uint32_t, hipStream_t) { touch_sum(d, n, frames, bl); }
void launch_scale(const ScaleDesc* d, int n, uint32_t frames, uint32_t bl, int, hipStream_t) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) {
        touch_w(d[i].buf, (size_t)frames * sizeof(float2));
        touch(d[i].peaks, (size_t)((frames + bl - 1) / bl) * 4);
        touch(d[i].init_copy, 8);
        touch_w(d[i].state, sizeof(NormState));
        if (d[i].qmode) touch_w(d[i].pcm, (size_t)frames * 2 * (d[i].qmode == 1u ? 2 : 4));
    }
}
void launch_norm_fix(const SumDesc* d, int n, uint32_t frames, uint32_t bl, hipStream_t) { touch_sum(d, n, frames, bl); }
void launch_quantise(const QuantDesc* d, int n, uint32_t frames, hipStream_t) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) { touch(d[i].in, (size_t)frames * sizeof(float2)); touch_w(d[i].pcm, (size_t)frames * 2 * (d[i].qmode == 1u ? 2 : 4)); }
}
void launch_sample_loop(const LoopDesc* d, int n, uint32_t frames, hipStream_t) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) { touch(d[i].sample, ((size_t)d[i].len + 15) * sizeof(float2)); touch_w(d[i].out, (size_t)frames * sizeof(float2)); }
}
void launch_sample_multi(const MultiDesc* d, int n, uint32_t frames, hipStream_t) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) {
        touch(d[i].hits, (size_t)d[i].n_hits * sizeof(MultiHit)); touch_w(d[i].out, (size_t)frames * sizeof(float2));
        touch(d[i].tile_first, (size_t)((frames + kTileFrames - 1) / kTileFrames) * 4);
    }
}
void launch_sample_lerp(const LerpDesc* d, int n, uint32_t frames, hipStream_t) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) {
        touch(d[i].hits, (size_t)d[i].n_hits * sizeof(LerpHit)); touch_w(d[i].out, (size_t)frames * sizeof(float2));
        touch(d[i].tile_first, (size_t)((frames + kTileFrames - 1) / kTileFrames) * 4);
    }
}
void launch_debug_sine(const SineDesc* d, int n, uint32_t frames, uint32_t, hipStream_t) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) { touch_interval_tab(d[i].tab, frames); touch_w(d[i].out, (size_t)frames * sizeof(float2)); }
}
void launch_synth(const SynthDesc* d, int n, uint32_t frames, bool, hipStream_t) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) {
        touch_interval_tab(d[i].tab, frames); touch_w(d[i].out, (size_t)frames * sizeof(float2));
        const IntervalTab& t = d[i].tab;
        const size_t nv = t.ivoff[t.n_int];
        if (d[i].affine) {   // four records per voice, read as ONE 64-byte load a voice ahead: a spare voice behind the last; the host's liveness bits
            touch(t.voices, (nv + 1) * 4 * sizeof(float4));
            if (d[i].exact_sin) abort();
            for (size_t v = 0; v < nv; ++v) {
                uint32_t live; memcpy(&live, &t.voices[4 * v].z, 4);
                for (int o = 0; o < 3; ++o) {
                    const float4& r = t.voices[4 * v + 1 + o];
                    if (((live >> o) & 1u) != (uint32_t)!(r.z == 0.0f && r.w == 0.0f)) abort();
                }
                if (live >> 3) abort();
            }
        } else {
            touch(t.voices, nv * sizeof(float4));
        }
    }
}
void launch_sampsyn(const SampsynDesc* d, int n, uint32_t frames, hipStream_t) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) {
        touch_interval_tab(d[i].tab, frames); touch_w(d[i].out, (size_t)frames * sizeof(float2));
        touch(d[i].wt.quads, (size_t)d[i].wt.n_frames * d[i].wt.frame_len * sizeof(float4));
    }
}
int launch_sources(const SourceParts& P, uint32_t frames, void* zero, size_t zero_bytes, hipStream_t s) {
    if (zero && zero_bytes) memset(zero, 0, zero_bytes);
    if (P.n_synth) launch_synth(P.synth, P.n_synth, frames, true, s);
    if (P.n_sampsyn) launch_sampsyn(P.sampsyn, P.n_sampsyn, frames, s);
    if (P.n_lerp) launch_sample_lerp(P.lerp, P.n_lerp, frames, s);
    if (P.n_env) launch_adsr_env(P.env, P.n_env, frames, s);
    return 1;
}
void launch_adsr(const AdsrVDesc* d, int n, uint32_t frames, uint32_t, hipStream_t) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) { touch_interval_tab(d[i].tab, frames); touch_terms(d[i].ins, d[i].k, frames); touch_w(d[i].out, (size_t)frames * sizeof(float2)); }
}
void launch_band_pass(const BandDesc* d, int n, uint32_t frames, hipStream_t) {
    touch_descs(d, n);
    for (int i = 0; i < n; ++i) { touch_terms(d[i].ins, d[i].k, frames); touch_w(d[i].out, (size_t)frames * sizeof(float2)); touch_w(d[i].state, sizeof(BandState)); }
}

}  // namespace tdk
