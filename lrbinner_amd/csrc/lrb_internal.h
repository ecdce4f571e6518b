// lrb_internal.h -- shared between the two translation units of liblrb_hip.so
#ifndef LRB_INTERNAL_H
#define LRB_INTERNAL_H

// printf-style with exactly two %s slots; stores the thread-local message that
// lrb_last_error() returns.
void lrb_set_error(const char *fmt, const char *a, const char *b);

#endif
