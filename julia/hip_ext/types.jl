# New compute modes (pattern: src/core/types.jl:29-35).  `adback` is kept for signature parity with the other
# MatrixMode types; the pullback / pushforward of the Dense chain is hand-written in the HIP kernels.
abstract type HIPMatrixMode{ADBack} <: MatrixMode{ADBack} end

"Hutchinson with vector-Jacobian products (the estimator of `LuxVecJacMatrixMode`, src/core/utils.jl:150-159)."
struct HIPVecJacMatrixMode{ADBack <: ADTypes.AbstractADType} <: HIPMatrixMode{ADBack}
    adback::ADBack
end

"Hutchinson with Jacobian-vector products (the estimator of `LuxJacVecMatrixMode`, src/core/utils.jl:161-170)."
struct HIPJacVecMatrixMode{ADBack <: ADTypes.AbstractADType} <: HIPMatrixMode{ADBack}
    adback::ADBack
end

HIPVecJacMatrixMode() = HIPVecJacMatrixMode(ADTypes.AutoZygote())
HIPJacVecMatrixMode() = HIPJacVecMatrixMode(ADTypes.AutoZygote())
