// base/Time.hpp -- minimal stand-in for Rock's base::Time (base-types is not available in this image).
// Only what UnscentedKalmanFilter.hpp uses (src/UnscentedKalmanFilter.hpp:9,30,86,93 of the
// reference): an int64 microsecond count with isNull(), toSeconds(), operator- and factories.
// When the real base-types package is on the include path, drop this directory from -I.
#ifndef UKFB_BASE_TIME_SHIM_HPP
#define UKFB_BASE_TIME_SHIM_HPP

#include <stdint.h>

namespace base {

struct Time {
    int64_t microseconds;
    Time() : microseconds(0) {}
    static Time fromMicroseconds(int64_t us) { Time t; t.microseconds = us; return t; }
    static Time fromSeconds(double s) { Time t; t.microseconds = static_cast<int64_t>(s * 1000000.0); return t; }
    bool isNull() const { return microseconds == 0; }
    double toSeconds() const { return static_cast<double>(microseconds) / 1000000.0; }
    int64_t toMicroseconds() const { return microseconds; }
    Time operator-(const Time& o) const { return fromMicroseconds(microseconds - o.microseconds); }
    Time operator+(const Time& o) const { return fromMicroseconds(microseconds + o.microseconds); }
    bool operator==(const Time& o) const { return microseconds == o.microseconds; }
    bool operator<(const Time& o) const { return microseconds < o.microseconds; }
};

}  // namespace base

#endif
