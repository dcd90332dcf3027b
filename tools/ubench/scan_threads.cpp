// Host-only micro-benchmark: T threads scan the same MP3 file into private, preallocated sinks.  Does one scan slow
// another down?  g++ -O3 -std=c++17 -pthread -I<csrc> -I<include> scan_threads.cpp <host sources> -o scan_threads
//   usage: scan_threads file.mp3 [iters]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "mp3s_host.h"
using namespace mp3s;

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + k);
    fclose(f);
    const int iters = argc > 2 ? atoi(argv[2]) : 50;
    for (int T : {1, 2, 3, 4, 6}) {
        std::vector<double> ms(T);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                std::vector<uint8_t> blob(file.size() * 2 + 4096);
                std::vector<mp3s_frame_side> side(file.size() / 96 + 16);
                std::vector<mp3s_frame_hdr> hdr(side.size());
                ParsedStream p;
                double best = 1e9, sum = 0;
                for (int i = 0; i < iters; i++) {
                    ScanSink s;
                    s.blob = blob.data(); s.blob_cap = blob.size(); s.side = side.data(); s.side_cap = side.size(); s.hdr = hdr.data(); s.lean = true;
                    const auto t0 = std::chrono::steady_clock::now();
                    const int rc = parse_stream_sink(file.data(), file.size(), p, &s);
                    const double d = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    if (rc) { fprintf(stderr, "rc %d\n", rc); exit(1); }
                    if (i) { sum += d; if (d < best) best = d; }
                }
                ms[t] = sum / (iters - 1);
            });
        for (auto &x : th) x.join();
        double a = 0;
        for (double x : ms) a += x;
        printf("threads %d: mean scan %.3f ms\n", T, a / T);
    }
    return 0;
}
