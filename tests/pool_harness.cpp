// tests/pool_harness.cpp -- the device-block pool (ilupp_amd/csrc/pool.h) on the CPU with a mock back end, built with
// -fsanitize=address,undefined by tests/test_pool.py (GPU sanitizers are not available on the target pool).
// Exit code 0 = every check held; the sanitizers abort on any misuse of the mock's memory.
#include <assert.h>
#include <stdio.h>
#include <string.h>

#include <random>
#include <set>
#include <vector>

#include "pool.h"

using namespace ilupp;

static std::set<void *> g_backend_live;
static size_t g_backend_bytes = 0, g_budget = (size_t)1 << 40;
static std::unordered_map<void *, size_t> g_sizes;
static int g_dev = 0;
static int g_double_backend_free = 0;

static int mock_alloc(void **p, size_t bytes)
{
    if (g_backend_bytes + bytes > g_budget) return 2;                 // "out of memory"
    *p = malloc(bytes);
    memset(*p, 0xab, bytes);
    g_backend_live.insert(*p); g_sizes[*p] = bytes; g_backend_bytes += bytes;
    return 0;
}
static int mock_release(void *p)
{
    if (!g_backend_live.count(p)) { ++g_double_backend_free; return 1; }
    g_backend_live.erase(p); g_backend_bytes -= g_sizes[p]; g_sizes.erase(p);
    free(p);
    return 0;
}
static int mock_device() { return g_dev; }

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "pool_harness: check failed at line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv)
{
    const bool expect_abort = argc > 1 && !strcmp(argv[1], "double-release-strict");
    {
        BlockPool P(PoolBackend{mock_alloc, mock_release, mock_device}, 1 << 20);
        void *a = nullptr, *b = nullptr, *c = nullptr;
        CHECK(P.acquire(&a, 1000) == 0 && P.acquire(&b, 1000) == 0);
        CHECK(a != b && P.live_blocks() == 2);
        memset(a, 1, 1000); memset(b, 2, 1000);
        CHECK(P.release(a) == 0 && P.cached() == 1024 && P.kept_blocks() == 1);
        // the kept block comes back for the same size on the same device, not for another device
        g_dev = 1;
        CHECK(P.acquire(&c, 1000) == 0 && c != a);
        g_dev = 0;
        void *d = nullptr;
        CHECK(P.acquire(&d, 1000) == 0 && d == a && P.cached() == 0);
        // a second release of a block is an error, never ignored -- also while the block sits in the kept list
        CHECK(P.release(d) == 0);
        if (expect_abort) { (void)P.release(d); return 0; }            // (ILUPP_POOL_STRICT=1: must abort here)
        CHECK(P.release(d) == BlockPool::kNotLive && P.bad_releases() == 1);
        // ... and after it has been handed to a new owner the stale release must not take it away from that owner
        void *e = nullptr;
        CHECK(P.acquire(&e, 1000) == 0 && e == a);
        CHECK(P.release(b) == 0);
        CHECK(P.release(b) == BlockPool::kNotLive);
        CHECK(P.is_live(e));
        int stack = 0;
        CHECK(P.release(&stack) == BlockPool::kNotLive);                  // a pointer the pool never handed out
        CHECK(P.release(e) == 0 && P.release(c) == 0 && P.release(nullptr) == 0);
        CHECK(P.live_blocks() == 0);
        // the limit: the oldest kept blocks leave first
        P.trim();
        CHECK(P.cached() == 0 && g_backend_live.empty());
        std::vector<void *> v(8);
        for (auto &q : v) CHECK(P.acquire(&q, 256 * 1024) == 0);
        for (auto &q : v) CHECK(P.release(q) == 0);
        CHECK(P.cached() <= (1u << 20) && P.kept_blocks() == 4 && g_backend_live.size() == 4);
        CHECK(g_backend_live.count(v[7]) && !g_backend_live.count(v[0]));
        P.set_limit(256 * 1024);
        CHECK(P.kept_blocks() == 1 && g_backend_live.size() == 1 && g_backend_live.count(v[7]));
        void *big = nullptr;
        CHECK(P.acquire(&big, 4 << 20) == 0 && P.release(big) == 0);     // larger than the limit: straight back
        CHECK(g_backend_live.size() == 1);
        // a failing back-end allocation gives the cache back and retries
        P.set_limit(1 << 30);
        for (auto &q : v) CHECK(P.acquire(&q, 1 << 20) == 0);
        for (auto &q : v) CHECK(P.release(q) == 0);
        g_budget = g_backend_bytes + (1 << 20);
        void *x = nullptr;
        CHECK(P.acquire(&x, 6 << 20) == 0 && P.cached() == 0);
        CHECK(P.release(x) == 0);
        g_budget = (size_t)1 << 40;
        // sizes from 1 MiB on are served from eight buckets per power of two: a slightly smaller request meets the kept block
        CHECK(BlockPool::bucket(1000) == 1024 && BlockPool::bucket((1 << 20) + 1) == (1 << 20) + (1 << 17) && BlockPool::bucket(6 << 20) == (6 << 20));
        CHECK(BlockPool::bucket(((size_t)937 << 20) + 12345) == ((size_t)960 << 20));
        {
            void *y = nullptr, *z = nullptr;
            CHECK(P.acquire(&y, (5 << 20) - 4096) == 0 && P.release(y) == 0);
            CHECK(P.acquire(&z, (5 << 20) - 300000) == 0 && z == y && P.release(z) == 0);
            P.trim();
        }
        // owner lanes (a batch of constructions on several streams): a kept block goes back only to the owner that freed it -- or, once
        // that owner has synchronised and disowned its blocks, to everybody; everybody's kept blocks serve every owner
        {
            P.trim();
            void *o1 = nullptr, *o2 = nullptr, *q = nullptr;
            CHECK(P.acquire(&o1, 3000, 1) == 0 && P.acquire(&o2, 3000, 2) == 0 && o1 != o2);
            CHECK(P.release(o1, 1) == 0 && P.kept_blocks() == 1);
            CHECK(P.acquire(&q, 3000, 2) == 0 && q != o1);                 // owner 2 does not get owner 1's unsynchronised block
            CHECK(P.release(q, 2) == 0);
            void *r = nullptr;
            CHECK(P.acquire(&r, 3000, 0) == 0 && r != o1 && r != q);      // nor does "everybody"
            CHECK(P.release(r, 0) == 0);
            void *s1 = nullptr;
            CHECK(P.acquire(&s1, 3000, 1) == 0 && s1 == o1);               // its own owner does
            CHECK(P.release(s1, 1) == 0);
            void *t2 = nullptr;
            CHECK(P.acquire(&t2, 3000, 2) == 0 && t2 == q);                // (owner 2 finds its own kept block first)
            CHECK(P.release(t2, 2) == 0);
            P.disown(1);
            void *u = nullptr, *u2 = nullptr;
            CHECK(P.acquire(&u, 3000, 3) == 0 && (u == o1 || u == r));     // disowned blocks and everybody's serve a third owner
            CHECK(P.acquire(&u2, 3000, 3) == 0 && (u2 == o1 || u2 == r) && u2 != u);
            CHECK(P.release(u, 3) == 0 && P.release(u2, 3) == 0 && P.release(o2, 2) == 0);
            P.disown(2); P.disown(3);
            const size_t kept = P.kept_blocks();
            P.set_limit(1);                                                 // the age list still knows every block after the re-tagging
            CHECK(P.kept_blocks() == 0 && kept == 4 && g_backend_live.empty());
            P.set_limit(1 << 30);
        }
        // random traffic
        std::mt19937 rng(7);
        std::vector<std::pair<void *, size_t>> held;
        for (int it = 0; it < 20000; ++it) {
            if (held.empty() || (rng() % 3 && held.size() < 200)) {
                const size_t bytes = (size_t)256 << (rng() % 8);
                void *q = nullptr;
                CHECK(P.acquire(&q, bytes) == 0);
                memset(q, (int)(it & 255), bytes);
                held.push_back({q, bytes});
            } else {
                const size_t i = rng() % held.size();
                CHECK(P.release(held[i].first) == 0);
                held[i] = held.back(); held.pop_back();
            }
            if (it % 997 == 0) P.set_limit((size_t)(rng() % 64) << 10);
            if (it % 4099 == 0) P.trim();
        }
        for (auto &h : held) CHECK(P.release(h.first) == 0);
        CHECK(P.live_blocks() == 0);
    }
    CHECK(g_backend_live.empty() && g_backend_bytes == 0 && g_double_backend_free == 0);
    printf("pool_harness: ok\n");
    return 0;
}
