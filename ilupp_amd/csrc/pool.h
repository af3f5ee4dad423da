// ilupp_amd/csrc/pool.h -- the size-keyed pool of device blocks behind pool_malloc / pool_free (api.hip), as a host-only class over
// an allocation back end, so that it can be exercised on the CPU under AddressSanitizer with a mock back end (tests/test_pool.py
// builds tests/pool_harness.cpp; GPU sanitizers are not available on the target pool).
//
// hipMalloc / hipFree of multi-GB blocks cost milliseconds and serialise the device; a factorisation that is repeated (time stepping,
// re-factorisation, the benchmark loop) asks for the same sizes every time, so freed blocks are KEPT, keyed by (device, size), and
// handed out again.  Rules:
//   * a block is either LIVE (handed out) or KEPT (freed, cached) or gone (returned to the back end): release() of anything that
//     is not live is an error -- reported, never ignored (a second free of a block that has meanwhile been handed to a new owner
//     would recycle memory in use); ILUPP_POOL_STRICT=1 aborts on it;
//   * the kept bytes are bounded by a limit the caller can set (ilupp_hip_set_cache_limit; default 24 GiB, ILUPP_CACHE_LIMIT_MB),
//     the oldest kept blocks go back to the back end first; a failing back-end allocation gives every kept block back and retries;
//   * the pool knows nothing of streams: a kept block may still be in use by kernels queued on the stream of whoever freed it.  Work on
//     ONE stream is safe by stream order; for callers that work on several streams at once (the batched construction: one worker per
//     matrix) every acquire / release names an OWNER: a kept block is handed out again only to the owner that freed it, until that
//     owner has synchronised its stream and given its kept blocks to everybody (disown).  Owner 0 is "everybody".
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <list>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>

namespace ilupp {

struct PoolBackend {
    int (*alloc)(void **p, size_t bytes);      // 0 = success
    int (*release)(void *p);                   // 0 = success
    int (*device)();                           // the current device
};

class BlockPool {
public:
    enum { kOk = 0, kNotLive = -1001, kBackend = -1002 };

    explicit BlockPool(const PoolBackend &b, size_t limit_bytes) : be_(b), limit_(limit_bytes)
    {
        const char *s = getenv("ILUPP_POOL_STRICT");
        strict_ = s && *s && *s != '0';
    }
    ~BlockPool() { trim(); }

    // the size a request is served with: multiples of 256 bytes; from 1 MiB on, one of eight sizes per power of two (at most 12.5 % more) --
    // requests that differ a little (the levels of a multilevel factorisation, each a bit smaller than the one before) then meet the
    // same kept blocks instead of sending a GB-sized block back to the driver and asking it for another one
    static size_t bucket(size_t bytes)
    {
        if (bytes == 0) bytes = 16;
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes < ((size_t)1 << 20)) return bytes;
        int k = 0;
        while (((size_t)2 << k) <= bytes) ++k;                  // 2^k <= bytes < 2^(k+1)
        const size_t step = (size_t)1 << (k - 3);
        return (bytes + step - 1) & ~(step - 1);
    }

    // a block of at least `bytes`: a kept one of the same device and bucket, else a new one
    int acquire(void **p, size_t bytes, int owner = 0)
    {
        bytes = bucket(bytes);
        const int dev = be_.device();
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto it = kept_.find(Key(dev, bytes, owner));
            if ((it == kept_.end() || it->second.empty()) && owner != 0) it = kept_.find(Key(dev, bytes, 0));
            if (it != kept_.end() && !it->second.empty()) {
                const Kept k = it->second.back();
                it->second.pop_back();
                if (it->second.empty()) kept_.erase(it);
                age_.erase(k.age);
                cached_ -= bytes;
                *p = k.p;
                live_[*p] = Key(dev, bytes, 0);
                ++hits_;
                return kOk;
            }
        }
        int e = be_.alloc(p, bytes);
        if (e != 0) {                 // out of memory: give the cache back and retry once
            trim();
            e = be_.alloc(p, bytes);
        }
        if (e != 0) return e;
        std::lock_guard<std::mutex> lk(mu_);
        live_[*p] = Key(dev, bytes, 0);
        ++misses_;
        return kOk;
    }

    // back to the pool (kept while the limit allows, the oldest kept blocks leave first)
    int release(void *p, int owner = 0)
    {
        if (!p) return kOk;
        std::list<void *> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto it = live_.find(p);
            if (it == live_.end()) {
                ++bad_releases_;
                if (strict_) {
                    fprintf(stderr, "[ilupp] pool: release of %p, which is not a live block of the pool (freed twice, or never handed out)\n", p);
                    abort();
                }
                return kNotLive;
            }
            const Key key(it->second.dev, it->second.bytes, owner);
            live_.erase(it);
            if (key.bytes <= limit_) {
                age_.push_back(AgeEntry(key, p));
                auto last = age_.end(); --last;
                kept_[key].push_back(Kept{p, last});
                cached_ += key.bytes;
                // over the limit: the oldest kept blocks go
                while (cached_ > limit_ && !age_.empty()) {
                    const AgeEntry a = age_.front();
                    remove_kept(a.first, a.second);
                    drop.push_back(a.second);
                }
            } else {
                drop.push_back(p);
            }
        }
        int rc = kOk;
        for (void *q : drop)
            if (be_.release(q) != 0) rc = kBackend;
        return rc;
    }

    // the kept blocks of `owner` become everybody's (the owner has synchronised the stream it worked on)
    void disown(int owner)
    {
        if (owner == 0) return;
        std::lock_guard<std::mutex> lk(mu_);
        for (auto it = kept_.begin(); it != kept_.end();) {
            if (it->first.owner != owner) { ++it; continue; }
            std::list<Kept> &dst = kept_[Key(it->first.dev, it->first.bytes, 0)];
            for (Kept &k : it->second) { k.age->first.owner = 0; dst.push_back(k); }
            it = kept_.erase(it);
        }
    }

    // every kept block back to the back end
    void trim()
    {
        std::list<void *> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (const AgeEntry &a : age_) drop.push_back(a.second);
            age_.clear(); kept_.clear(); cached_ = 0;
        }
        for (void *q : drop) (void)be_.release(q);
    }

    void set_limit(size_t bytes)
    {
        std::list<void *> drop;
        {
            std::lock_guard<std::mutex> lk(mu_);
            limit_ = bytes;
            while (cached_ > limit_ && !age_.empty()) {
                const AgeEntry a = age_.front();
                remove_kept(a.first, a.second);
                drop.push_back(a.second);
            }
        }
        for (void *q : drop) (void)be_.release(q);
    }
    size_t limit() const { return limit_; }
    size_t cached() const { std::lock_guard<std::mutex> lk(mu_); return cached_; }
    size_t live_blocks() const { std::lock_guard<std::mutex> lk(mu_); return live_.size(); }
    size_t kept_blocks() const { std::lock_guard<std::mutex> lk(mu_); return age_.size(); }
    size_t bad_releases() const { std::lock_guard<std::mutex> lk(mu_); return bad_releases_; }
    size_t hits() const { return hits_; }
    size_t misses() const { return misses_; }
    bool is_live(void *p) const { std::lock_guard<std::mutex> lk(mu_); return live_.count(p) != 0; }

private:
    struct Key {                                             // (device, bytes, owner)
        int dev; size_t bytes; int owner;
        Key() : dev(0), bytes(0), owner(0) {}
        Key(int d, size_t b, int o) : dev(d), bytes(b), owner(o) {}
        bool operator<(const Key &o) const { return dev != o.dev ? dev < o.dev : bytes != o.bytes ? bytes < o.bytes : owner < o.owner; }
    };
    typedef std::pair<Key, void *> AgeEntry;
    struct Kept { void *p; std::list<AgeEntry>::iterator age; };

    void remove_kept(const Key &key, void *p)                // (lock held)
    {
        auto it = kept_.find(key);
        if (it == kept_.end()) return;
        for (auto k = it->second.begin(); k != it->second.end(); ++k)
            if (k->p == p) {
                age_.erase(k->age);
                it->second.erase(k);
                cached_ -= key.bytes;
                break;
            }
        if (it->second.empty()) kept_.erase(it);
    }

    PoolBackend be_;
    mutable std::mutex mu_;
    std::map<Key, std::list<Kept>> kept_;                    // kept blocks by (device, size), the youngest last
    std::list<AgeEntry> age_;                                // all kept blocks, the oldest first
    std::unordered_map<void *, Key> live_;
    size_t cached_ = 0, limit_ = 0, hits_ = 0, misses_ = 0, bad_releases_ = 0;
    bool strict_ = false;
};

}  // namespace ilupp
