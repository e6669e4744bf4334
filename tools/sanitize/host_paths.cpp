// host_paths.cpp -- drives the device-free code of libjsnoop_gpu under ASan / UBSan (tools/sanitize/run_sanitizers.sh):
//   * jsnoop_selftest_tables: the decode-table builders of the parallel path on random canonical Huffman table sets;
//   * the JFIF front end (js_jfif_walk) and the descriptor / geometry / table-resolution code (js_describe_image) on synthetic
//     files and on thousands of header / scan mutations of them (truncations, flipped bytes, hostile segment lengths).
// No HIP device is touched: decoders are built directly from the library's host classes.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../jpegsnoop_amd/csrc/jsnoop_host.h"
#include "../../oracle/jpeg_synth.h"

static uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: host_paths <libjsnoop_synth.so>\n"); return 2; }
    void* so = dlopen(argv[1], RTLD_NOW);
    if (!so) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    typedef size_t (*enc_fn)(const JsynthParams*, uint8_t*, size_t);
    enc_fn encode = (enc_fn)dlsym(so, "jsynth_encode");
    if (!encode) { fprintf(stderr, "jsynth_encode missing\n"); return 2; }

    int bad = jsnoop_selftest_tables(1, 300);
    printf("selftest_tables: %d disagreements over 300 table sets\n", bad);
    const int badb = jsnoop_selftest_bytes(5, 5000);               // the sixteen-bytes-per-step marker searches of the staging code (unaligned loads up to the last byte)
    printf("selftest_bytes: %d disagreements over 5000 buffers\n", badb);
    bad += badb;

    size_t walks = 0, accepted = 0, described = 0;
    std::vector<uint8_t> buf(1 << 20);
    for (int c = 0; c < 24; c++) {
        JsynthParams p; memset(&p, 0, sizeof p);
        p.width = 16 + rnd() % 300; p.height = 16 + rnd() % 200; p.hs = 1 + rnd() % 2; p.vs = 1 + rnd() % 2; p.quality = 20 + rnd() % 80;
        p.restart_interval = (rnd() % 3) ? 0 : 1 + rnd() % 9; p.gray = (rnd() % 5) == 0; p.optimize_huffman = rnd() % 2; p.progressive = (rnd() % 6) == 0 ? 1 + rnd() % 2 : 0;
        p.noise_sigma = 12; p.seed = 100 + c;
        size_t n = encode(&p, buf.data(), buf.size());
        if (n > buf.size()) { buf.resize(n); n = encode(&p, buf.data(), buf.size()); }
        std::vector<uint8_t> file(buf.begin(), buf.begin() + n);
        for (int m = 0; m < 400; m++) {
            std::vector<uint8_t> f = file;
            const int kind = m == 0 ? 0 : 1 + rnd() % 5;
            const size_t hdr = std::min<size_t>(f.size(), 700);
            if (kind == 1) f.resize(1 + rnd() % f.size());                                            // truncation
            else if (kind == 2) for (int k = 0; k < 1 + (int)(rnd() % 4); k++) f[rnd() % hdr] ^= (uint8_t)(1u << (rnd() % 8));   // bit flips in the header
            else if (kind == 3) f[rnd() % hdr] = (uint8_t)rnd();                                      // a random header byte
            else if (kind == 4) { const size_t at = rnd() % hdr; f[at] = 0xFF; if (at + 1 < f.size()) f[at + 1] = (uint8_t)(0xC0 + rnd() % 0x30); }   // a stray marker
            else if (kind == 5) { const size_t at = rnd() % hdr; if (at + 1 < f.size()) { f[at] = 0xFF; f[at + 1] = 0xFF; } }                    // fill bytes
            JsnoopDecoder d; unsigned start = 0;
            walks++;
            if (js_jfif_walk(&d, f.data(), f.size(), &start) == 0) {
                accepted++;
                JsImage im; JsTableSet* ts = new JsTableSet;
                if (js_describe_image(&d, &im, ts, (uint32_t)f.size(), start, 1, 1)) described++;
                delete ts;
            }
        }
    }
    printf("jfif_walk: %zu files (24 synthetic x 400 mutations), %zu accepted, %zu described\n", walks, accepted, described);
    printf("host paths done: %s\n", bad == 0 ? "clean" : "SELFTEST DISAGREEMENTS");
    return bad == 0 ? 0 : 1;
}
