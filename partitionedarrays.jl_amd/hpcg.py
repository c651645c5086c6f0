"""HPCG's conjugate-gradient loop on the device path (identity preconditioner).

Mirrors /root/reference/HPCG/src/ref_cg.jl:40-134 (`cg_iterator!`, `iterate`, `ref_cg!`) and
HPCG/src/hpcg_utils.jl:6-17 (`mul_no_lat!`).  The multigrid preconditioner (`Pl`) is out of scope;
with `Pl = Identity()` the loop is BASELINE config 4: per iteration one mul!, two dots + one norm, three axpy-like
broadcasts, all on device-resident PVectors.
"""
from __future__ import annotations

from .p_sparse_matrix import mul_, mul_no_overlap_
from .p_vector import axpby_, copy_, dot, norm, similar

mul_no_lat_ = mul_no_overlap_     # HPCG/src/hpcg_utils.jl:6-17: blocking consistent!, then the local product


def ref_cg_(x, A, b, maxiter=50, tolerance=0.0, overlap=True, history=None):
    """ref_cg!(x,A,b,timing_data;tolerance,maxiter,Pl=Identity()) -> x, residual0, residual, iters.
    `overlap=True` uses mul! (latency hiding, src/p_sparse_matrix.jl:2090); False uses mul_no_lat! as HPCG does."""
    mv = mul_ if overlap else mul_no_lat_
    # cg_iterator! (ref_cg.jl:76-96).  Vectors that are multiplied by A live on the column partition (= row
    # partition + ghosts, HPCG/src/sparse_matrix.jl:119).
    u = similar(x)                       # u .= 0
    r = similar(x)
    c = similar(x)
    copy_(r, b)                          # copyto!(r,b)
    mv(c, A, x)                          # c = A*x
    axpby_(r, -1.0, c, 1.0)              # r .-= c
    residual0 = residual = norm(r)
    rho = 1.0
    iters = 0
    while not (iters >= maxiter or residual / residual0 <= tolerance):      # done(it,iteration) (:23)
        copy_(c, r)                      # ldiv!(c, Identity(), r)  (:48)
        rho_prev = rho
        rho = dot(c, r)                  # (:52)
        beta = rho / rho_prev
        axpby_(u, 1.0, c, beta)          # u .= c .+ beta .* u     (:56)
        mv(c, A, u)                      # c = A*u                 (:59)
        uc = dot(u, c)                   # (:60)
        alpha = rho / uc
        axpby_(x, alpha, u, 1.0)         # x .+= alpha .* u        (:64)
        axpby_(r, -alpha, c, 1.0)        # r .-= alpha .* c        (:65)
        residual = norm(r)               # (:67)
        iters += 1
        if history is not None:
            history.append(residual)
    return x, residual0, residual, iters
