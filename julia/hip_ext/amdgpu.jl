# Optional: include after `import AMDGPU` (or load as a package extension with AMDGPU as a weak dependency) when
# `icnf.device = MLDataDevices.AMDGPUDevice()`.  ROCArrays are then passed to the library by device pointer, on the task's
# HIP stream, with no staging copies.
is_device_array(::AMDGPU.ROCArray) = true
current_stream(::AMDGPU.ROCArray) = Ptr{Cvoid}(UInt(AMDGPU.stream().stream.handle))
current_device_id() = Int32(AMDGPU.device_id(AMDGPU.device()) - 1)
