// Sanitizer run of the host engine on random projects (built by tests/test_compile_asan.py with
// g++ -fsanitize=address,undefined against tests/mock_hip.cpp -- no GPU, nothing computed): every project goes through the
// real front-end and C ABI -- State::refresh, render, scan_exact + render, a render that continues, block pulls -- in the
// three band modes, un-chunked and in 4 096-frame chunks, so that the event compiler and the descriptor / arena builder of
// csrc/engine.cpp run over what tests/test_gpu_fuzz.py's generator makes with every write bounds-checked.
//   usage: asan_compile <dir> ...     each <dir> holds project.lua and meta.txt ("<buffer length>")
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "termdaw_amd.h"

static std::string slurp(const std::string& p) {
    std::string s; FILE* f = fopen(p.c_str(), "rb"); if (!f) return s;
    char b[4096]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) s.append(b, n); fclose(f); return s;
}
int main(int argc, char** argv) {
    size_t renders = 0, rejected = 0, failed = 0;
    for (int a = 1; a < argc; ++a) {
        const std::string dir = argv[a];
        const std::string lua = slurp(dir + "/project.lua");
        const size_t bl = (size_t)atol(slurp(dir + "/meta.txt").c_str());
        if (lua.empty() || !bl) { fprintf(stderr, "bad project dir %s\n", dir.c_str()); return 2; }
        for (int mode = 0; mode < 3; ++mode)
            for (int chunked = 0; chunked < 2; ++chunked) {
                td_state* s = td_state_new("", 48000, bl);
                if (!s) return 3;
                td_state_set_option(s, "band_mode", mode);
                if (chunked) td_state_set_option(s, "max_chunk_frames", 4096);
                if (mode == 2 && chunked) td_state_set_option(s, "band_guard_ppb", 0);   // (every audited render is done again: the redo path)
                if (!td_state_refresh_source(s, lua.c_str())) { ++rejected; td_state_free(s); continue; }
                std::vector<unsigned char> pcm(td_state_render_to_memory(s, nullptr, 0) + 16);
                for (int k = 0; k < 3; ++k) {
                    if (k == 1 && !td_state_scan_exact(s)) ++failed;
                    if (pcm.size() > 16 && !td_state_render_to_memory(s, pcm.data(), pcm.size())) ++failed;
                    ++renders;
                }
                std::vector<float> l(bl), r(bl);
                for (int k = 0; k < 3; ++k) {   // Graph::render block pulls (graph.rs:182-193) behind the renders
                    if (td_graph_render_block(td_state_graph(s), td_state_samplebank(s), td_state_flowwbank(s), l.data(), r.data()) < 0) ++failed;
                    td_flowwbank_set_time_to_next_block(td_state_flowwbank(s));
                }
                td_state_free(s);
            }
    }
    {   // the N > 1 exchange on host memory (td_comm_init_host): two accepted projects as one rank's batch of a two-rank job
        std::vector<td_state*> st;
        for (int a = 1; a < argc && st.size() < 2; ++a) {
            const std::string dir = argv[a];
            td_state* s = td_state_new("", 48000, (size_t)atol(slurp(dir + "/meta.txt").c_str()));
            if (s && td_state_refresh_source(s, slurp(dir + "/project.lua").c_str())) st.push_back(s);
            else if (s) td_state_free(s);
        }
        if (st.size() == 2) {
            td_batch* b = td_batch_new();
            for (td_state* s : st) if (td_batch_add(b, td_state_graph(s), td_state_samplebank(s), td_state_flowwbank(s)) < 0) ++failed;
            struct Ctx { size_t calls, n; } ctx{0, 0};
            td_comm* c = td_comm_init_host([](void* p, float* table, size_t n) -> int {
                Ctx* x = (Ctx*)p; x->calls += 1; x->n = n;
                for (size_t i = 0; i < n; ++i) table[i] = table[i] > 0.5f ? table[i] : 0.5f;   // ("the other rank" holds 0.5 everywhere)
                return 1; }, &ctx, 1, 2);
            if (!c) ++failed;
            td_batch_rewind(b);
            if (!td_batch_render_all(b, 3, 16)) ++failed;
            if (c && !td_batch_exchange_peaks(b, c, 3)) ++failed;
            if (!td_batch_sync(b)) ++failed;
            float table[6] = {0};
            if (!td_batch_read_peak_table(b, table, 6)) ++failed;
            if (ctx.calls != 1 || ctx.n != 6) ++failed;
            for (float v : table) if (!(v >= 0.5f)) ++failed;
            td_comm_free(c);
            td_batch_free(b);
        }
        for (td_state* s : st) td_state_free(s);
    }
    printf("asan_compile done: %d projects, %zu renders, %zu rejected refreshes, %zu failed calls\n", argc - 1, renders, rejected, failed);
    return failed ? 1 : 0;
}
