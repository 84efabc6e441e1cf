// Sanitizer driver for the header-only host classes that need no device (tests/test_sanitizers.py; built by
// tests/sanitize/Makefile with -fsanitize=address,undefined).  TEST INFRASTRUCTURE.
//   host_classes_asan selftest <tmpdir>   encoders against hand-assembled proto3 bytes, encode -> decode round trips,
//                                          an EXR file written and its header re-read
//   host_classes_asan fuzz <seed> <cases> readScatterSample on mutated / random records: every case must end in a decoded
//                                          record or a C++ exception, never in a sanitizer report
#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdexcept>
#include <string>

#include "../../deepestscatter_amd/host/Collectors.h"
#include "../../deepestscatter_amd/host/Exr.h"

using namespace DeepestScatter;

static int fail(const char* what)
{
    std::fprintf(stderr, "host_classes_driver: %s\n", what);
    return 1;
}

static int selftest(const std::string& dir)
{
    // Result { light_intensity = 1.5, is_converged = true }: 0d 00 00 c0 3f 10 01
    const std::string r = Persistance::result(1.5f, true);
    if (r != std::string("\x0d\x00\x00\xc0\x3f\x10\x01", 7)) return fail("Result bytes");
    if (!Persistance::result(0.f, false).empty()) return fail("proto3 default values must be omitted");
    const float p[3] = { 0.25f, -1.0f, 3.0f }, d[3] = { 0.f, 1.f, 0.f };
    for (int id : { 0, 1, 300, -7, 2047 * 2048 })
    {
        const std::string s = Persistance::scatterSample(id, p, d);
        float p2[3], d2[3];
        Persistance::readScatterSample(s, p2, d2);
        for (int k = 0; k < 3; k++)
            if (p2[k] != p[k] || d2[k] != d[k]) return fail("ScatterSample round trip");
    }
    const float l[3] = { -0.03f, -0.25f, 0.8f };
    if (Persistance::sceneSetup("clouds/a.vdb", 7000.f, l).size() != 2 + 12 + 5 + 2 + 15) return fail("SceneSetup size");
    std::vector<uint8_t> grid(2250, 7);
    if (Persistance::disneyDescriptor(grid.data(), grid.size()).size() != 1 + 2 + 2250) return fail("DisneyDescriptor size");
    // EXR: R, G, B FLOAT scan lines (Camera.cpp:154-174); header magic and size of the file
    const uint32_t w = 7, h = 5;
    std::vector<float> rgba((size_t)w * h * 4);
    for (size_t i = 0; i < rgba.size(); i++) rgba[i] = (float)i * 0.5f;
    const std::string path = dir + "/t.exr";
    Exr::writeRgbFloat(path, w, h, rgba.data());
    std::ifstream f(path, std::ios::binary);
    std::vector<char> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (bytes.size() < 8 || std::memcmp(bytes.data(), "\x76\x2f\x31\x01", 4) != 0) return fail("EXR magic");
    if (bytes.size() < (size_t)w * h * 12) return fail("EXR payload");
    std::puts("selftest ok");
    return 0;
}

static int fuzz(uint32_t seed, uint32_t cases)
{
    std::mt19937 rng(seed);
    const float p[3] = { 0.25f, -1.0f, 3.0f }, d[3] = { 0.5f, 1.f, 0.f };
    uint32_t decoded = 0, rejected = 0;
    for (uint32_t c = 0; c < cases; c++)
    {
        std::string s = Persistance::scatterSample((int32_t)(rng() % 5000) - 10, p, d);
        switch (rng() % 4)
        {
        case 0: s.resize(rng() % (s.size() + 1)); break;                                       // truncated
        case 1: for (int k = 0, n = 1 + rng() % 4; k < n; k++) s[rng() % s.size()] ^= (char)(1u << (rng() % 8)); break;   // bit flips
        case 2: { const size_t at = rng() % s.size(); s[at] = (char)(rng() & 0xff); break; }   // one byte replaced (a length or a tag)
        default: s.assign(rng() % 40, '\0'); for (auto& ch : s) ch = (char)(rng() & 0xff); break;   // noise
        }
        float p2[3], d2[3];
        try
        {
            Persistance::readScatterSample(s, p2, d2);
            decoded++;
        }
        catch (const std::exception&)
        {
            rejected++;
        }
    }
    std::printf("fuzz: %u decoded, %u rejected\n", decoded, rejected);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 3 && std::string(argv[1]) == "selftest") return selftest(argv[2]);
    if (argc >= 4 && std::string(argv[1]) == "fuzz") return fuzz((uint32_t)std::atoi(argv[2]), (uint32_t)std::atoi(argv[3]));
    std::fprintf(stderr, "usage: host_classes_asan selftest <tmpdir> | fuzz <seed> <cases>\n");
    return 2;
}
