import numpy as np, sys
sys.path.insert(0, "/root/repo")
from muscle_synergies_amd.synth import emg_matrix
from muscle_synergies_amd.init import initialize_nmf, nndsvd_init_batched
for (T, m, k) in [(2000, 16, 5), (500, 8, 3), (77, 6, 6)]:
    Xs = np.stack([np.ascontiguousarray(emg_matrix(500 + b, T=T, m=m, k_true=min(5, m), dtype=np.float64)) for b in range(2)])
    W0, H0 = nndsvd_init_batched(Xs, k, init="nndsvd")
    W0, H0 = W0.cpu().numpy(), H0.cpu().numpy()
    for b in range(2):
        Wr, Hr = initialize_nmf(Xs[b], k, init="nndsvd", random_state=0)
        U, S, Vt = np.linalg.svd(Xs[b], full_matrices=False)
        print(T, m, k, b, "max|dH|", np.abs(H0[b]-Hr).max(), "max|dW|", np.abs(W0[b]-Wr).max(), "per-col dW", np.abs(W0[b]-Wr).max(axis=0), "S", S[:k+1])
