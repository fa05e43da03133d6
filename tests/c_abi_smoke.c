/* Plain-C consumer of include/termdaw_amd.h: proves the header is valid C (no C++-isms) and that the
 * library links and runs its host-side entry points without a GPU.  Built and run by
 * tests/test_host_logic.py::test_c_abi_from_plain_c. */
#include <stdio.h>
#include <string.h>

#include "termdaw_amd.h"

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            printf("FAILED line %d: %s [%s]\n", __LINE__, #cond, td_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(void) {
    td_graph* g = td_graph_new(1024, 48000);
    td_flowwbank* fb = td_flowwbank_new(48000, 1024);
    td_event ev[2] = {{0.5f, 60.0f, 0.9f}, {0.75f, 60.0f, 0.0f}};
    float adsr[6] = {0.01f, 0.1f, 0.8f, 0.1f, 0.2f, 0.01f};
    CHECK(g && fb);
    CHECK(td_flowwbank_add_events(fb, "notes", ev, 2) == 0);
    CHECK(td_flowwbank_get_index(fb, "notes") == 0 && td_flowwbank_get_index(fb, "nope") == -1);
    CHECK(td_graph_add_debug_sine(g, "sine", 1.0f, 0.0f, 0));
    CHECK(td_graph_add_adsr(g, "env", 1.0f, 0.0f, 1.0f, 0, 0, 1, -1, adsr, 6));
    CHECK(!td_graph_add_adsr(g, "bad", 1.0f, 0.0f, 1.0f, 0, 0, 1, -1, adsr, 5));
    CHECK(td_graph_add_normalize(g, "out", 1.0f, 0.0f));
    CHECK(td_graph_connect(g, "sine", "env") && td_graph_connect(g, "env", "out"));
    CHECK(!td_graph_connect(g, "out", "env"));   /* cycle */
    CHECK(!td_graph_check(g));                   /* no output yet */
    CHECK(td_graph_set_output(g, "out") && td_graph_check(g));
    CHECK(td_graph_vertex_count(g) == 3);
    td_graph_reset_normalize_vertices(g);
    CHECK(td_graph_get_normalization_value(g, "out") > 0.0f && td_graph_get_normalization_value(g, "sine") == -1.0f);
    td_graph_set_time(g, 2048);
    CHECK(td_graph_get_time(g) == 2048);
    CHECK(td_graph_set_option(g, "fuse_sources", 0) && !td_graph_set_option(g, "no_such_option", 1));
    if (td_device_count() == 0) {   /* no CPU fallback: rendering must fail loudly */
        td_samplebank* sb = td_samplebank_new(48000);
        CHECK(td_graph_render_all(g, sb, fb, 4, 16) == 0);
        CHECK(strstr(td_last_error(), "no HIP device") != NULL);
        td_samplebank_free(sb);
    }
    td_graph_free(g);
    td_flowwbank_free(fb);
    printf("c abi ok\n");
    return 0;
}
