// adsr_math.h -- piecewise-linear envelope evaluators shared by host (event compiler) and device
// (kernels).  Follows /root/reference/src/adsr.rs:1-114: IEEE f32, same operation order.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TD_HD __host__ __device__ __forceinline__
#else
#define TD_HD inline
#endif
#include <math.h>

namespace tdk {

struct AdsrConfD {   // AdsrConf adsr.rs:1-12
    float std_vel, attack_sec, attack_vel, decay_sec, decay_vel, sustain_sec, sustain_vel, release_sec,
        release_vel;
};

TD_HD float lerpf(float a, float b, float t) { return a + t * (b - a); }   // adsr.rs:41-44

TD_HD float ads_internal(const AdsrConfD& c, float t) {   // adsr.rs:46-60
    if (t <= c.attack_sec) return lerpf(c.std_vel, c.attack_vel, t / c.attack_sec);
    if (t <= c.attack_sec + c.decay_sec)
        return lerpf(c.attack_vel, c.decay_vel, (t - c.attack_sec) / c.decay_sec);
    if (t <= c.attack_sec + c.decay_sec + c.sustain_sec)
        return lerpf(c.decay_vel, c.sustain_vel, (t - c.attack_sec - c.decay_sec) / c.sustain_sec);
    return -1000.0f;
}
TD_HD float apply_ads(const AdsrConfD& c, float t) {   // adsr.rs:62-69
    float res = ads_internal(c, t);
    return res <= -1.0f ? c.sustain_vel : res;
}
TD_HD float apply_r(const AdsrConfD& c, float t, float old_val) {   // adsr.rs:71-73
    return lerpf(old_val, c.release_vel, fminf(t / c.release_sec, 1.0f));
}
TD_HD float apply_adsr(const AdsrConfD& c, float t) {   // adsr.rs:75-86
    float res = ads_internal(c, t);
    if (res <= -1.0f)
        return lerpf(c.sustain_vel, c.release_vel,
                     fminf((t - c.attack_sec - c.decay_sec - c.sustain_sec) / c.release_sec, 1.0f));
    return res;
}
TD_HD float apply_r_rt(const AdsrConfD& c, float t, float rt) {   // adsr.rs:89-92
    return apply_r(c, t, apply_ads(c, rt));
}
TD_HD float adsr_max_vel(const AdsrConfD& c) {   // adsr.rs:32-38
    return fmaxf(fmaxf(fmaxf(fmaxf(c.std_vel, c.attack_vel), c.decay_vel), c.sustain_vel), c.release_vel);
}

}  // namespace tdk
