// Stand-in for the oneTBB 2022.1.0 headers the reference includes (not installed): serial equivalents of
// blocked_range / parallel_for / parallel_reduce / concurrent_vector / global_control / this_task_arena.  One
// thread, ranges visited front to back: a deterministic instance of what the reference's TBB code may do.
// TEST INFRASTRUCTURE.
#pragma once
#include <cstddef>
#include <vector>

namespace tbb {
template <class It>
class blocked_range {
    It b_, e_;

public:
    using const_iterator = It;
    blocked_range(It b, It e) : b_(b), e_(e) {}
    It begin() const { return b_; }
    It end() const { return e_; }
};
template <class Range, class Body>
void parallel_for(const Range &r, const Body &body) {
    body(r);
}
template <class Range, class Value, class Body, class Reduction>
Value parallel_reduce(const Range &r, const Value &identity, const Body &body, const Reduction &) {
    return body(r, identity);
}
template <class T>
using concurrent_vector = std::vector<T>;
struct global_control {
    enum parameter { max_allowed_parallelism };
    global_control(parameter, size_t) {}
};
namespace this_task_arena {
inline int max_concurrency() { return 1; }
}  // namespace this_task_arena
}  // namespace tbb
