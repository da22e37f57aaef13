/* TEST INFRASTRUCTURE (part of the CPU oracle; never linked into the product).
 * patches[s, n] of a celeste_problem_t in either of its two forms (include/celeste_mi355x.h): the dense
 * [s * n_images + n] table, or the sparse list sorted by (source, image) in which every pair that is not listed
 * is the reference's empty clamp_box patch (imaged_sources.jl:10-14). */
#ifndef CELESTE_ORACLE_PATCH_LOOKUP_H
#define CELESTE_ORACLE_PATCH_LOOKUP_H
#include "../include/celeste_mi355x.h"

static const celeste_patch_t *oracle_patch_at(const celeste_problem_t *pr, int s, int n) {
    /* an empty patch still carries a PSF (unused: it covers no pixel) */
    static const double unit_psf[8 * 6] = {1, 0, 0, 1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 0, 0, 1, 0, 1,
                                           1, 0, 0, 1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 0, 0, 1, 0, 1};
    static const celeste_patch_t empty = {0, 0, 0, 0, 0, {1, 0, 0, 1}, {0, 0}, {0, 0}, unit_psf, 0, 0};
    if (pr->n_patch_entries <= 0) return &pr->patches[(size_t)s * pr->n_images + n];
    const int64_t key = (int64_t)s * pr->n_images + n;
    int64_t lo = 0, hi = pr->n_patch_entries;
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        const int64_t k = (int64_t)pr->patch_source[mid] * pr->n_images + pr->patch_image[mid];
        if (k < key) lo = mid + 1; else hi = mid;
    }
    if (lo < pr->n_patch_entries && (int64_t)pr->patch_source[lo] * pr->n_images + pr->patch_image[lo] == key)
        return &pr->patches[lo];
    return &empty;
}
#endif
