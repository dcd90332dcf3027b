// Container formats and message framing on the host: WAV header in/out, "<count>#<text>" framing
// and the reveal parse.  Byte shuffling only -- nothing here is on the device path.
#include <stdio.h>
#include <string.h>

#include "mp3s_host.h"

namespace mp3s {
namespace {

long find4(const uint8_t *buf, size_t n, const char *tag)
{
    for (size_t i = 0; i + 4 <= n; i++)
        if (!memcmp(buf + i, tag, 4)) return (long)i;
    return -1;
}

// struct.unpack on a slice of the 128-byte header buffer: a short slice is a struct.error in the reference
bool le(const uint8_t *buf, size_t n, long at, int bytes, uint32_t *v)
{
    if (at < 0 || (size_t)at + bytes > n) return false;
    uint32_t x = 0;
    for (int i = 0; i < bytes; i++) x |= (uint32_t)buf[at + i] << (8 * i);
    *v = x;
    return true;
}

bool is_space(uint8_t c)   // what int() strips below code point 256: C isspace plus NEL and NBSP (not 0x1c..0x1f)
{
    return (c >= 0x09 && c <= 0x0d) || c == 0x20 || c == 0x85 || c == 0xa0;
}

// int(text) of CPython for a string of code points < 256: surrounding whitespace, one sign, decimal digits with single
// underscores between them.  Saturates at +-2^62 (a length that large only decides "longer than the text").
bool py_int(const uint8_t *s, size_t n, int64_t *out)
{
    size_t a = 0, b = n;
    while (a < b && is_space(s[a])) a++;
    while (b > a && is_space(s[b - 1])) b--;
    bool neg = false;
    if (a < b && (s[a] == '+' || s[a] == '-')) neg = s[a++] == '-';
    if (a == b) return false;
    int64_t v = 0;
    bool prev_digit = false;
    for (size_t i = a; i < b; i++) {
        if (s[i] == '_') {
            if (!prev_digit || i + 1 == b) return false;
            prev_digit = false;
            continue;
        }
        if (s[i] < '0' || s[i] > '9') return false;
        if (v < ((int64_t)1 << 58)) v = v * 10 + (s[i] - '0');
        else v = (int64_t)1 << 62;
        prev_digit = true;
    }
    *out = neg ? -v : v;
    return true;
}

}  // namespace

// reference encoder/WAV_Reader.py:30-107 (+ check_bitrate_index :109-111).  Returns MP3S_OK, MP3S_E_EXIT with the
// reference's sys.exit text in *msg, or MP3S_E_MALFORMED where the reference dies in struct.unpack / a division.
int wav_parse(const uint8_t *file, size_t len, int bitrate_kbps, mp3s_wav_info *o, const char **msg)
{
    const size_t n = len < 128 ? len : 128;   // the header is searched in the first 128 bytes only
    uint32_t v = 0;
    *msg = "Bad WAVE file.";
    long idx = find4(file, n, "RIFF");
    if (idx < 0) return MP3S_E_EXIT;
    *msg = "short WAVE header";
    if (!le(file, n, idx + 4, 4, &v)) return MP3S_E_MALFORMED;
    *msg = "Bad WAVE file.";
    if (find4(file, n, "WAVE") < 0) return MP3S_E_EXIT;
    idx = find4(file, n, "fmt ");
    if (idx < 0) return MP3S_E_EXIT;
    idx += 4;
    *msg = "short WAVE header";
    if (!le(file, n, idx, 4, &v)) return MP3S_E_MALFORMED;
    *msg = "Unsupported WAVE file, compression used instead of PCM.";
    if (v != 16) return MP3S_E_EXIT;
    idx += 4;
    if (!le(file, n, idx, 2, &v)) { *msg = "short WAVE header"; return MP3S_E_MALFORMED; }
    if (v != 1) return MP3S_E_EXIT;
    idx += 2;
    *msg = "short WAVE header";
    if (!le(file, n, idx, 2, &v)) return MP3S_E_MALFORMED;
    o->channels = (int32_t)v;
    idx += 2;
    if (!le(file, n, idx, 4, &v)) return MP3S_E_MALFORMED;
    o->samplerate = (int32_t)v;
    *msg = "Unsupported sampling frequency.";
    if (v != 32000 && v != 44100 && v != 48000) return MP3S_E_EXIT;
    idx += 4 + 4 + 2;   // byte rate and block align are read but never used
    *msg = "short WAVE header";
    if (!le(file, n, idx, 2, &v)) return MP3S_E_MALFORMED;
    o->bits_per_sample = (int32_t)v;
    *msg = "Unsupported WAVE file, samples not int8, int16 or int32 type.";
    if (v != 8 && v != 16 && v != 32) return MP3S_E_EXIT;
    *msg = "Bad WAVE file.";
    idx = find4(file, n, "data");
    if (idx < 0) return MP3S_E_EXIT;
    idx += 4;
    *msg = "short WAVE header";
    if (!le(file, n, idx, 4, &v)) return MP3S_E_MALFORMED;
    *msg = "WAVE header with zero channels (ZeroDivisionError in the reference)";
    if (o->channels == 0) return MP3S_E_MALFORMED;
    // int(sub_chunk2_size * 8 / bits_per_sample / channels): two float divisions, truncation
    o->num_of_samples = (int64_t)((double)((uint64_t)v * 8) / (double)o->bits_per_sample / (double)o->channels);
    o->data_offset = idx + 4;
    // np.fromfile(f, 'int16', num_of_samples * channels * 2): always int16, up to TWICE the declared values, cut by EOF
    const int64_t want = o->num_of_samples * o->channels * 2;
    const int64_t have = (size_t)o->data_offset <= len ? (int64_t)((len - (size_t)o->data_offset) / 2) : 0;
    o->n_values = want < have ? want : have;
    o->bitrate = bitrate_kbps;
    int sri, bri, whole;
    *msg = "Unsupported bitrate configuration.";
    if (stream_params(o->samplerate, bitrate_kbps, &sri, &bri, &whole)) return MP3S_E_EXIT;
    *msg = "";
    return MP3S_OK;
}

// the 44 bytes scipy.io.wavfile.write puts in front of int16 data (reference MP3_Parser.py:86-93 -> scipy)
void wav_header(int64_t n_rows, int nch, int rate, uint8_t *h)
{
    auto put = [&](int at, uint32_t v, int bytes) { for (int i = 0; i < bytes; i++) h[at + i] = (uint8_t)(v >> (8 * i)); };
    const uint32_t nb = (uint32_t)(n_rows * nch * 2);
    memcpy(h, "RIFF", 4); put(4, 36 + nb, 4); memcpy(h + 8, "WAVEfmt ", 8);
    put(16, 16, 4); put(20, 1, 2); put(22, (uint32_t)nch, 2); put(24, (uint32_t)rate, 4);
    put(28, (uint32_t)(rate * nch * 2), 4); put(32, (uint32_t)(nch * 2), 2); put(34, 16, 2);
    memcpy(h + 36, "data", 4); put(40, nb, 4);
}

// reference steganography.py:10-24, 42-50: str(len(message)) + '#' + message, UTF-8, MSB first.  len() counts code
// points while the payload is UTF-8 bytes, so a non-ASCII message reveals truncated -- kept.
void message_frame(const uint8_t *utf8, size_t n, std::vector<uint8_t> &bits)
{
    size_t chars = 0;
    for (size_t i = 0; i < n; i++) chars += (utf8[i] & 0xc0) != 0x80;
    char head[32];
    const int hn = snprintf(head, sizeof head, "%zu#", chars);
    bits.clear();
    bits.reserve((hn + n) * 8);
    auto push = [&](uint8_t b) { for (int k = 7; k >= 0; k--) bits.push_back((b >> k) & 1); };
    for (int i = 0; i < hn; i++) push((uint8_t)head[i]);
    for (size_t i = 0; i < n; i++) push(utf8[i]);
}

// reference decoder/decoder.py:90-108: bytes -> chr() each, digits up to the first '#', int() or 0, slice, UTF-8.
void message_reveal(const uint8_t *bits, size_t n_bits, std::vector<uint8_t> &text)
{
    const size_t n = n_bits / 8;   // zip(*[iter(bits)] * 8) drops an incomplete last byte
    std::vector<uint8_t> s(n);
    for (size_t i = 0; i < n; i++) {
        uint8_t b = 0;
        for (int k = 0; k < 8; k++) b = (uint8_t)((b << 1) | (bits[i * 8 + k] & 1));
        s[i] = b;
    }
    size_t head = 0;
    while (head < n && s[head] != '#') head++;   // no '#': the whole string is the "length"
    int64_t mlen = 0;
    if (!py_int(s.data(), head, &mlen)) { mlen = 0; head = 0; }
    // Python slices: [a:] or [a:a+mlen] with negative ends counted from the back and everything clamped
    const int64_t L = (int64_t)n, a = (int64_t)head + 1;
    int64_t lo = a < L ? a : L, hi = L;
    if (!(a + mlen > L)) {
        hi = a + mlen;
        if (hi < 0) { hi += L; if (hi < 0) hi = 0; }
    }
    text.clear();
    for (int64_t i = lo; i < hi; i++) {   // bytes(str, 'utf-8') of code points below 256
        if (s[i] < 0x80) text.push_back(s[i]);
        else { text.push_back((uint8_t)(0xc0 | (s[i] >> 6))); text.push_back((uint8_t)(0x80 | (s[i] & 0x3f))); }
    }
}

}  // namespace mp3s
