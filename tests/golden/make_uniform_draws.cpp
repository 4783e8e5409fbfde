// Prints the draw of cactus_sample_f16 (std::mt19937(seed), std::uniform_real_distribution<float>(0, 1))
// for a few seeds as "seed hex-bits-of-the-float"; built with this image's g++ (libstdc++).
//   g++ -O1 -o /tmp/draws tests/golden/make_uniform_draws.cpp && /tmp/draws > tests/golden/uniform_draws.txt
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
int main() {
    const uint32_t seeds[] = {1u, 2u, 42u, 5489u, 123456789u, 0x80000000u, 0xFFFFFFFFu, 1700000000u, 7u, 99991u,
                              20240611u, 3141592653u};
    for (uint32_t s : seeds) {
        std::mt19937 gen(s);
        std::uniform_real_distribution<float> dist(0.0f, 1.0f);
        const float v = dist(gen);
        uint32_t b;
        std::memcpy(&b, &v, 4);
        std::printf("%u %08x\n", s, b);
    }
    // seeds whose first output rounds to 2^32 as a float (the result is clamped below 1)
    uint32_t found = 0;
    for (uint32_t s = 1; s < 4000000u && found < 2; s++) {
        std::mt19937 gen(s);
        if (gen() >= 0xFFFFFF80u) {
            std::mt19937 g2(s);
            std::uniform_real_distribution<float> dist(0.0f, 1.0f);
            const float v = dist(g2);
            uint32_t b;
            std::memcpy(&b, &v, 4);
            std::printf("%u %08x\n", s, b);
            found++;
        }
    }
    return 0;
}
