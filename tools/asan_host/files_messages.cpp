#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include "mp3s_host.h"
using namespace mp3s;
// Host-side file / message code under ASan+UBSan: exact-size heap buffers, random and structured inputs.
int main(int argc, char **argv) {
    const int scale = argc > 1 ? atoi(argv[1]) : 100;   // percent of the full iteration counts
    std::mt19937_64 rng(7);
    // message_reveal on random / structured bit strings
    long nrev = 0;
    for (int it = 0; it < 2000 * scale; it++) {
        size_t n = rng() % 400;
        uint8_t *bits = (uint8_t *)malloc(n ? n : 1);
        int mode = it % 4;
        std::vector<uint8_t> text;
        if (mode == 0) for (size_t i = 0; i < n; i++) bits[i] = rng() & 1;
        else {
            // "<digits/ws/sign/_>#<junk>" as bits
            const char alphabet[] = "0123456789 _+-#\t\x85\xa0" "a#";
            size_t nb = n / 8; std::vector<uint8_t> s(nb);
            for (auto &c : s) c = alphabet[rng() % (sizeof alphabet - 1)];
            for (size_t i = 0; i < nb * 8; i++) bits[i] = (s[i / 8] >> (7 - i % 8)) & 1;
            for (size_t i = nb * 8; i < n; i++) bits[i] = rng() & 1;
        }
        message_reveal(bits, n, text); nrev += text.size();
        free(bits);
    }
    // message_frame
    for (int it = 0; it < 200 * scale; it++) {
        size_t n = rng() % 300; uint8_t *s = (uint8_t *)malloc(n ? n : 1);
        for (size_t i = 0; i < n; i++) s[i] = rng();
        std::vector<uint8_t> bits; message_frame(s, n, bits); free(s);
    }
    // wav_parse on mutated headers
    long okw = 0;
    for (int it = 0; it < 3000 * scale; it++) {
        uint8_t h[44]; wav_header(rng() % 5000, 1 + rng() % 2, (int[]){44100, 48000, 32000, 22050}[rng() % 4], h);
        size_t extra = rng() % 300; size_t n = 44 + extra;
        if (it % 5 == 0) n = rng() % 60;
        uint8_t *buf = (uint8_t *)malloc(n ? n : 1);
        for (size_t i = 0; i < n; i++) buf[i] = i < 44 ? h[i] : (uint8_t)rng();
        int muts = rng() % 4;
        for (int m = 0; m < muts && n; m++) buf[rng() % (n < 48 ? n : 48)] = rng();
        if (it % 11 == 0 && n > 50) { size_t at = rng() % 40; memmove(buf + at + 6, buf + at, n - at - 6); }
        mp3s_wav_info w; const char *msg = nullptr;
        int rc = wav_parse(buf, n, (int[]){128, 320, 64, 0, -1, 77}[rng() % 6], &w, &msg);
        if (rc == 0) {
            okw++;
            // what the API then does with the result: touch the data range it names
            volatile uint8_t sink = 0;
            if (w.n_values) { sink ^= buf[w.data_offset]; sink ^= buf[w.data_offset + 2 * w.n_values - 1]; }
        }
        free(buf);
    }
    printf("reveal bytes %ld, wav ok %ld\n", nrev, okw);
}
