// Host parser / scanner / reveal / WAV parse over every file of a directory, under ASan+UBSan; each file is copied to
// an exact-size heap block so that an overread of one byte is reported.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <dirent.h>
#include "mp3s_host.h"
using namespace mp3s;
static std::vector<uint8_t> slurp(const std::string &p) {
    std::vector<uint8_t> v; FILE *f = fopen(p.c_str(), "rb"); if (!f) return v;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (n) fread(v.data(), 1, n, f); fclose(f); return v;
}
int main(int argc, char **argv) {
    DIR *d = opendir(argv[1]); int n = 0, ok = 0, walked = 0;
    while (dirent *e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        // exact-size heap copy so that any overread trips ASan
        std::vector<uint8_t> v = slurp(std::string(argv[1]) + "/" + e->d_name);
        uint8_t *buf = (uint8_t *)malloc(v.size() ? v.size() : 1); if (v.size()) memcpy(buf, v.data(), v.size());
        ParsedStream p; ScannedStream s;
        int rc = parse_stream(buf, v.size(), p, nullptr);
        int rc2 = parse_stream(buf, v.size(), p, &s);
        ok += rc == 0;
        if (rc2 == 0 && s.gpu_ok) {   // the one-frame decode the device pipeline falls back to, on every scanned frame
            std::vector<int16_t> is(2304);
            mp3s_granule_si si4[4];
            for (size_t f = 0; f < s.side.size(); f++) parse_scanned_frame(s.side[f], s.blob.data(), is.data(), si4);
        }
        // the frame walk (mp3s_walk_stream / the pipe's chunks): in pieces of 1, 7 and "all" frames, with exact-size ref and
        // table blocks; a walk that calls itself regular must give the scan's frames
        for (long piece : {1L, 7L, 1L << 20}) {
            FrameWalker w;
            if (w.open(buf, v.size())) continue;
            w.tables_wanted = piece == 7 ? 40 : 0x7fffffffffffffffL;
            std::vector<FrameRef> all;
            while (!w.ended && !w.irregular) {
                const long cap = piece < (long)(v.size() / 24 + 16) ? piece : (long)(v.size() / 24 + 16);
                FrameRef *r = (FrameRef *)malloc(cap * sizeof(FrameRef)); uint8_t *t4 = (uint8_t *)calloc(cap, 4);
                const long got = w.next(r, cap, t4, 0, 0);
                all.insert(all.end(), r, r + got);
                free(r); free(t4);
                if (got == 0 && !w.ended && !w.irregular) { printf("walk stalls on %s\n", e->d_name); return 1; }
            }
            if (w.irregular || all.empty()) continue;
            walked++;
            if (rc2 != 0 || all.size() != s.side.size()) { printf("walk disagrees with the scan on %s (%zu vs %zu, rc %d)\n", e->d_name, all.size(), s.side.size(), rc2); return 1; }
            for (size_t f = 0; f < all.size(); f++)
                if (all[f].md_off != s.side[f].md_off || all[f].md_len != s.side[f].md_len) { printf("walk: frame %zu of %s\n", f, e->d_name); return 1; }
            uint16_t hist[9];
            for (size_t f = 0; f < all.size(); f += all.size() / 5 + 1) FrameWalker::history(all.data(), (long)f, hist);
            std::vector<int16_t> is(2304); mp3s_granule_si si4[4]; bool alone = false;
            w.decode_last(is.data(), si4, &alone);
        }
        {   // stego bits from per-frame table words: any words, any carry
            std::vector<uint64_t> tsel(v.size() / 8);
            if (!tsel.empty()) memcpy(tsel.data(), buf, tsel.size() * 8);
            for (int nch = 1; nch <= 2; nch++) { uint8_t carry[4] = {1, 31, 0, 7}; std::vector<uint8_t> bits; stego_bits_from_tsel(tsel.data(), (long)tsel.size(), nch, carry, bits); }
        }
        std::vector<uint8_t> text; message_reveal(p.bits.data(), p.bits.size(), text);
        mp3s_wav_info w; const char *msg; wav_parse(buf, v.size(), 128, &w, &msg);
        free(buf); n++;
    }
    closedir(d);
    printf("files %d parsed ok %d walked %d\n", n, ok, walked);
    return 0;
}
