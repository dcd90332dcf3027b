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
    DIR *d = opendir(argv[1]); int n = 0, ok = 0;
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
        std::vector<uint8_t> text; message_reveal(p.bits.data(), p.bits.size(), text);
        mp3s_wav_info w; const char *msg; wav_parse(buf, v.size(), 128, &w, &msg);
        free(buf); n++;
    }
    closedir(d);
    printf("files %d parsed ok %d\n", n, ok);
    return 0;
}
