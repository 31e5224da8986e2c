// The host side of the stamped blocks (slam-tricks_amd/csrc/common.hpp): a block is accepted only if EVERY 64-byte line carries the
// awaited stamp and a check word that fits what was read -- a line caught half-written (new stamp, old payload; or the other way
// round) must be refused.  The lines are packed here by the documented layout (six payload doubles | check = stamped_mix chain over stamp and payload | stamp), i.e. this also
// pins the layout the device code writes.  No device needed.
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../slam-tricks_amd/csrc/common.hpp"

static void pack(std::vector<double>& blk, const std::vector<double>& payload, double stamp) {
    const int n = (int)payload.size(), nl = stba::stamped_lines(n);
    blk.assign((size_t)8 * nl, 0.0);
    for (int L = 0; L < nl; ++L) {
        unsigned long long chk = stba::STAMPED_SALT, w;
        std::memcpy(&w, &stamp, 8); chk = stba::stamped_mix(chk, w);
        for (int q = 0; q < 6; ++q) {
            const double v = (L * 6 + q < n) ? payload[(size_t)L * 6 + q] : 0.0;
            blk[(size_t)8 * L + q] = v;
            std::memcpy(&w, &v, 8); chk = stba::stamped_mix(chk, w);
        }
        std::memcpy(&blk[(size_t)8 * L + 6], &chk, 8);
        blk[(size_t)8 * L + 7] = stamp;
    }
}

int main() {
    int bad = 0;
    auto expect = [&](bool cond, const char* what) { if (!cond) { std::printf("FAILED: %s\n", what); ++bad; } };
    for (int n : {1, 3, 6, 7, 9, 12, 13, 66}) {
        std::vector<double> pay((size_t)n), blk, old_blk, got((size_t)n, -1.0);
        for (int k = 0; k < n; ++k) pay[(size_t)k] = 0.25 * k - 3.0 + 1e-9 * k * k;
        pack(blk, pay, 41.0);
        auto is41 = [](double s) { return s == 41.0; };
        auto is42 = [](double s) { return s == 42.0; };
        auto ge40 = [](double s) { return s >= 40.0; };
        double st = 0.0;
        expect(stba::stamped_try_read(blk.data(), n, is41, got.data(), &st) && st == 41.0 && got == pay, "a complete block is read back");
        expect(!stba::stamped_try_read(blk.data(), n, is42, got.data()), "another stamp is not accepted");
        expect(stba::stamped_try_read(blk.data(), n, ge40, got.data()), "at-least acceptance");
        // the next hand-off half arrived: new stamp in a line whose payload is still the old one
        std::vector<double> pay2 = pay;
        for (auto& v : pay2) v += 1.0;
        std::vector<double> blk2;
        pack(blk2, pay2, 42.0);
        for (size_t line = 0; line < blk.size() / 8; ++line) {
            std::vector<double> torn = blk2;
            for (int q = 0; q < 6; ++q) torn[8 * line + q] = blk[8 * line + q];          // old payload, new check + stamp
            expect(!stba::stamped_try_read(torn.data(), n, is42, got.data()), "old payload under a new stamp is refused");
            torn = blk2;
            torn[8 * line + 6] = blk[8 * line + 6];                                        // old check word
            expect(!stba::stamped_try_read(torn.data(), n, is42, got.data()), "an old check word is refused");
            torn = blk2;
            torn[8 * line + 7] = 41.0;                                                     // a line that still carries the old stamp
            expect(!stba::stamped_try_read(torn.data(), n, is42, got.data()), "a line with the old stamp is refused");
            expect(!stba::stamped_try_read(torn.data(), n, ge40, got.data()) || blk.size() == 8, "lines with different stamps are refused");
        }
        std::vector<double> zeros(blk.size(), 0.0);
        expect(!stba::stamped_try_read(zeros.data(), n, ge40, got.data()) && !stba::stamped_try_read(zeros.data(), n, [](double) { return true; }, got.data()),
               "a zeroed block is never valid");
    }
    std::printf(bad ? "stamped_block FAILED %d\n" : "stamped_block ok\n", bad);
    return bad ? 1 : 0;
}
