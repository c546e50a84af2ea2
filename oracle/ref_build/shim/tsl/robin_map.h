// Stand-in for <tsl/robin_map.h> (tsl::robin_map 1.4.0 is not installed): the part of its interface the reference
// uses -- find / end / insert / erase(iterator) / contains / reserve / size / empty / clear / iteration, with
// it.value() for mutable access.  Iteration order is INSERTION order (the real container's is bucket order,
// which the reference never relies on); that is also the order the oracle defines for VoxelDownsample's output.
// TEST INFRASTRUCTURE.
#pragma once
#include <functional>
#include <list>
#include <unordered_map>
#include <utility>

namespace tsl {
template <class K, class V, class H = std::hash<K>>
class robin_map {
    using List = std::list<std::pair<const K, V>>;
    List items_;
    std::unordered_map<K, typename List::iterator, H> index_;

public:
    template <class It>
    struct iter : It {
        iter() = default;
        iter(It i) : It(i) {}
        const K &key() const { return (**this).first; }
        V &value() const { return const_cast<V &>((**this).second); }
    };
    using iterator = iter<typename List::iterator>;
    using const_iterator = iter<typename List::const_iterator>;
    iterator begin() { return items_.begin(); }
    iterator end() { return items_.end(); }
    const_iterator begin() const { return items_.begin(); }
    const_iterator end() const { return items_.end(); }
    const_iterator cbegin() const { return items_.cbegin(); }
    const_iterator cend() const { return items_.cend(); }
    iterator find(const K &k) {
        auto f = index_.find(k);
        return f == index_.end() ? items_.end() : f->second;
    }
    const_iterator find(const K &k) const {
        auto f = index_.find(k);
        return f == index_.end() ? items_.cend() : typename List::const_iterator(f->second);
    }
    bool contains(const K &k) const { return index_.find(k) != index_.end(); }
    std::pair<iterator, bool> insert(std::pair<K, V> kv) {
        auto f = index_.find(kv.first);
        if (f != index_.end()) return {iterator(f->second), false};
        items_.emplace_back(kv.first, std::move(kv.second));
        auto it = std::prev(items_.end());
        index_.emplace(kv.first, it);
        return {iterator(it), true};
    }
    iterator erase(iterator it) {
        index_.erase(it->first);
        return items_.erase(it);
    }
    void reserve(size_t n) { index_.reserve(n); }
    size_t size() const { return items_.size(); }
    bool empty() const { return items_.empty(); }
    void clear() {
        items_.clear();
        index_.clear();
    }
};
}  // namespace tsl
