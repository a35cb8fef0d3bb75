// Device-resident state of one balanced-BCE loss call (ocrs_models/train_detection.py:225-263), shared by the loss kernels (loss_optim.hip) and by the
// head backward that forms dL/dpred on the fly from it (det_bwd.hip: k_head_bwd_loss).
#pragma once
struct LossState {            // device-resident, one per loss call
    unsigned long long cnt[2];   // #pos, #neg
    unsigned long long k;        // min(cnt)
    unsigned prefix[2];          // radix-select prefix / final threshold bits per class
    unsigned long long need[2];  // how many still to take inside the current prefix bucket
    unsigned long long ties[2];  // elements equal to the threshold
    double sum_gt[2];            // sum of losses strictly above the threshold
    float loss;
    float frac[2];               // need / ties
    float inv2k;                 // 1 / (2k)
};

__device__ __forceinline__ unsigned loss_key(float v) { return __float_as_uint(v) & 0x7fffffffu; }

