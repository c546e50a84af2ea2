// Stand-in for <tsl/robin_map.h>.  tsl::robin_map 1.4.0 (3rdparty/tsl_robin/tsl_robin.cmake:24) is fetched from the network by
// the reference's build and is not installed here, so this header RESTATES its storage algorithm from the published
// sources -- as far as the reference can observe it, which is through ITERATION ORDER:
//   * VoxelDownsample emits its survivors by iterating the grid (core/VoxelUtils.cpp:17-19), and that order decides
//     which points AddPoints' order-dependent cap / spacing rule keeps (core/VoxelHashMap.cpp:98-118) and which point
//     of a 1.5 v voxel the second downsample keeps (pipeline/KissICP.cpp:72-73);
//   * Pointcloud() iterates the map (core/VoxelHashMap.cpp:72-81).
// What is restated (robin_hash.h of v1.4.0, with robin_map's default template arguments):
//   - open addressing over a power-of-two bucket array (rh::power_of_two_growth_policy<2>: bucket = hash & mask, growth x2),
//     default-constructed with 0 buckets, max_load_factor 0.5, min_load_factor 0 (no shrinking);
//   - insert: probe from the home bucket while dist <= bucket.dist (an empty bucket has dist -1); before placing,
//     grow (rehash to next_bucket_count) if size() >= load_threshold = size_t(float(bucket_count) * 0.5f) -- or a probe
//     sequence ran past DIST_FROM_IDEAL_BUCKET_LIMIT -- and probe again; then robin-hood placement: the carried element
//     takes a bucket whose occupant is STRICTLY closer to its own home, the occupant is carried on (equal distance:
//     the carried element moves on, so elements of one home bucket rotate when something is inserted in front of them);
//   - reserve(n) = rehash(ceil(float(n) / 0.5f)), rehash(c) = max(c, ceil(size / 0.5f)) rounded up to a power of two;
//   - rehash: a new array, the elements re-inserted in BUCKET ORDER of the old one;
//   - erase(iterator): clear + backward shift of the followers with dist > 0; returns the same bucket if the shift
//     brought an element into it, else the next occupied one;
//   - iteration: bucket 0 .. bucket_count - 1.
// NOT verified against the upstream sources (they are not on this machine): "parity unpinned" for this container, and
// DESIGN.md says so.  The arithmetic that matters for poses is only the ORDER; tests/naive_ref.py holds an independent
// Python statement of the same rules, and tests/test_ref_pins_oracle.py compares all three.
// TEST INFRASTRUCTURE.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <iterator>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

namespace tsl {
template <class K, class V, class H = std::hash<K>, class E = std::equal_to<K>>
class robin_map {
    static constexpr int kDistLimit = 8192;  // bucket_entry::DIST_FROM_IDEAL_BUCKET_LIMIT
    struct Bucket {
        int dist = -1;  // EMPTY_MARKER_DIST_FROM_IDEAL_BUCKET
        alignas(std::pair<K, V>) unsigned char raw[sizeof(std::pair<K, V>)];
        std::pair<K, V> &kv() { return *reinterpret_cast<std::pair<K, V> *>(raw); }
        const std::pair<K, V> &kv() const { return *reinterpret_cast<const std::pair<K, V> *>(raw); }
        bool empty() const { return dist < 0; }
        void set(int d, std::pair<K, V> &&v) {
            new (raw) std::pair<K, V>(std::move(v));
            dist = d;
        }
        void clear() {
            if (dist >= 0) kv().~pair();
            dist = -1;
        }
        Bucket() = default;
        Bucket(const Bucket &o) : dist(o.dist) {
            if (dist >= 0) new (raw) std::pair<K, V>(o.kv());
        }
        Bucket(Bucket &&o) noexcept : dist(o.dist) {
            if (dist >= 0) new (raw) std::pair<K, V>(std::move(o.kv()));
        }
        Bucket &operator=(const Bucket &) = delete;
        ~Bucket() { clear(); }
    };
    std::vector<Bucket> b_;
    size_t mask_ = 0, n_ = 0, threshold_ = 0;
    bool grow_next_ = false;
    H hash_;
    E eq_;

    static size_t pow2(size_t v) {
        if (v == 0) return 0;
        size_t p = 1;
        while (p < v) p <<= 1;
        return p;
    }
    size_t bucket_count() const { return b_.size(); }
    size_t next(size_t i) const { return (i + 1) & mask_; }
    void set_buckets(size_t count) {  // robin_hash(bucket_count, ...): the growth policy rounds up to a power of two
        count = pow2(count);
        std::vector<Bucket> nb(count);
        b_.swap(nb);
        mask_ = count ? count - 1 : 0;
        threshold_ = size_t(float(count) * 0.5f);
    }
    // insert_value_on_rehash / insert_value: carry `v` from bucket i at distance d until an empty bucket takes it
    void carry(size_t i, int d, std::pair<K, V> &&v, bool on_rehash) {
        for (;;) {
            Bucket &bk = b_[i];
            if (d > bk.dist) {
                if (bk.empty()) {
                    bk.set(d, std::move(v));
                    return;
                }
                if (!on_rehash && d > kDistLimit) grow_next_ = true;
                std::swap(v, bk.kv());
                std::swap(d, bk.dist);
            }
            ++d;
            i = next(i);
        }
    }
    void rehash_impl(size_t count) {
        std::vector<Bucket> old;
        old.swap(b_);
        set_buckets(count);
        for (Bucket &bk : old)  // bucket order of the old array
            if (!bk.empty()) carry(hash_(bk.kv().first) & mask_, 0, std::move(bk.kv()), true);
    }
    bool rehash_on_extreme_load(int dist) {
        if (grow_next_ || dist > kDistLimit || n_ >= threshold_) {
            rehash_impl(bucket_count() ? bucket_count() * 2 : 2);  // GrowthPolicy::next_bucket_count(): (mask + 1) * 2
            grow_next_ = false;
            return true;
        }
        return false;  // (min_load_factor is 0: m_try_shrink_on_next_insert never shrinks)
    }
    size_t find_bucket(const K &k) const {
        if (b_.empty()) return size_t(-1);
        size_t i = hash_(k) & mask_;
        int d = 0;
        while (d <= b_[i].dist) {
            if (eq_(b_[i].kv().first, k)) return i;
            i = next(i);
            ++d;
        }
        return size_t(-1);
    }

public:
    template <bool Const>
    class iter {
        friend class robin_map;
        using Map = typename std::conditional<Const, const robin_map, robin_map>::type;
        Map *m_ = nullptr;
        size_t i_ = 0;
        iter(Map *m, size_t i) : m_(m), i_(i) {}
        void skip() {
            while (i_ < m_->b_.size() && m_->b_[i_].empty()) ++i_;
        }

    public:
        iter() = default;
        iter(const iter<false> &o) : m_(o.m_), i_(o.i_) {}
        const std::pair<K, V> &operator*() const { return m_->b_[i_].kv(); }
        const std::pair<K, V> *operator->() const { return &m_->b_[i_].kv(); }
        const K &key() const { return m_->b_[i_].kv().first; }
        V &value() const { return const_cast<V &>(m_->b_[i_].kv().second); }
        iter &operator++() {
            ++i_;
            skip();
            return *this;
        }
        bool operator==(const iter &o) const { return i_ == o.i_; }
        bool operator!=(const iter &o) const { return i_ != o.i_; }
        using iterator_category = std::forward_iterator_tag;
        using value_type = std::pair<K, V>;
        using difference_type = std::ptrdiff_t;
        using pointer = const std::pair<K, V> *;
        using reference = const std::pair<K, V> &;
    };
    using iterator = iter<false>;
    using const_iterator = iter<true>;

    robin_map() = default;
    iterator begin() {
        iterator it(this, 0);
        it.skip();
        return it;
    }
    iterator end() { return iterator(this, b_.size()); }
    const_iterator begin() const {
        const_iterator it(this, 0);
        it.skip();
        return it;
    }
    const_iterator end() const { return const_iterator(this, b_.size()); }
    const_iterator cbegin() const { return begin(); }
    const_iterator cend() const { return end(); }

    iterator find(const K &k) {
        const size_t i = find_bucket(k);
        return i == size_t(-1) ? end() : iterator(this, i);
    }
    const_iterator find(const K &k) const {
        const size_t i = find_bucket(k);
        return i == size_t(-1) ? end() : const_iterator(this, i);
    }
    bool contains(const K &k) const { return find_bucket(k) != size_t(-1); }

    std::pair<iterator, bool> insert(std::pair<K, V> kv) {
        const size_t h = hash_(kv.first);
        size_t i = h & mask_;
        int d = 0;
        if (!b_.empty()) {
            while (d <= b_[i].dist) {
                if (eq_(b_[i].kv().first, kv.first)) return {iterator(this, i), false};
                i = next(i);
                ++d;
            }
        }
        while (rehash_on_extreme_load(d)) {
            i = h & mask_;
            d = 0;
            while (d <= b_[i].dist) {
                i = next(i);
                ++d;
            }
        }
        const size_t at = i;
        carry(i, d, std::move(kv), false);  // the new element takes bucket `at` (empty, or its occupant is carried on)
        ++n_;
        return {iterator(this, at), true};
    }

    iterator erase(iterator pos) {
        size_t prev = pos.i_;
        b_[prev].clear();
        --n_;
        size_t i = next(prev);
        while (b_[i].dist > 0) {  // backward shift
            b_[prev].set(b_[i].dist - 1, std::move(b_[i].kv()));
            b_[i].clear();
            prev = i;
            i = next(i);
        }
        if (b_[pos.i_].empty()) ++pos;
        return pos;
    }

    void reserve(size_t count) { rehash(size_t(std::ceil(float(count) / 0.5f))); }
    void rehash(size_t count) {
        const size_t need = size_t(std::ceil(float(n_) / 0.5f));
        rehash_impl(count > need ? count : need);
    }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    void clear() {  // min_load_factor 0: the bucket array is kept
        for (Bucket &bk : b_) bk.clear();
        n_ = 0;
        grow_next_ = false;
    }
};
}  // namespace tsl
