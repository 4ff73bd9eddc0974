#!/usr/bin/env python3
"""The reference tutorial's flow ("Finding muscle synergies") on the GPU engine, with synthetic raw EMG.

    raw EMG frame -> linear_envelope (or zero_center + rms) -> time_normalize -> normalize
                  -> find_synergies(df, 2, 6, solver='mu') -> VAF table -> pick the rank

Every call below has the reference's name and signature (`import muscle_synergies_amd as ms` instead of
`import muscle_synergies as ms`); the sample arithmetic runs on an MI355X.
"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd

import muscle_synergies_amd as ms
from muscle_synergies_amd.synth import raw_emg


def main(T=20_000, m=8, fs=2000, quiet=False):
    muscles = ["VL", "RF", "GMED", "TFL", "GMAXS", "GMAXI", "BF", "ST"][:m]
    raw = pd.DataFrame(raw_emg(1, T, m, fs=float(fs)), columns=muscles, index=np.arange(T) / fs)

    envelope = ms.linear_envelope(raw, critical_freqs=6, sampling_frequency=fs, order=4)      # zero-lag Butterworth
    envelope_rms = ms.rms(ms.zero_center(raw), window_size=0.1, sampling_frequency=fs)          # the RMS alternative
    processed = ms.normalize(ms.time_normalize(envelope.clip(lower=0), reduce_to=1000))

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        result = ms.find_synergies(processed, 2, 6, solver="mu", max_iter=20_000, tol=1e-6, random_state=0)
    vaf_all = result.vaf_values["All signals"]
    chosen = int(vaf_all[vaf_all >= 0.90].index[0]) if (vaf_all >= 0.90).any() else int(vaf_all.index[-1])
    if not quiet:
        print(result.vaf_values.round(4).to_string())
        print(f"smallest rank with VAF >= 0.90: {chosen}; iterations per rank: "
              f"{ {k: mdl.n_iter_ for k, mdl in result.model.items()} }")
        print(result.components[chosen].round(3).to_string())
    return raw, envelope, envelope_rms, processed, result, chosen


if __name__ == "__main__":
    main()
