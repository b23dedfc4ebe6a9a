# main_amdgpu.jl -- the reference's Julia driver (src/main.jl:1-66) bound to the HIP library.
# Same wrappers, same three index known-answer tests; ROCArray instead of CuArray.  The
# plotting sweeps of main.jl:85-163 are not reproduced.  Untested in the build image (no Julia).
using Libdl
using AMDGPU
using Test

get_ptr(A) = Base.unsafe_convert(Ptr{Cfloat}, A)
lib = Libdl.dlopen(joinpath(@__DIR__, "..", "conjugate-gradient_amd", "lib", "libconjugate_gradient_gpu.so"))

function add_jl(A::ROCArray{Float32}, B::ROCArray{Float32}, blocks, threads)
    sym = Libdl.dlsym(lib, :add_jl)
    @ccall $sym(get_ptr(A)::Ptr{Cfloat}, get_ptr(B)::Ptr{Cfloat}, blocks::Cint, threads::Cint)::Cvoid
end

function laplace_gpu_jl(ddf::ROCArray{Float32}, u::ROCArray{Float32}, dx::Cfloat, d, L, N, index_mode, blocks, threads)
    sym = Libdl.dlsym(lib, :laplace_gpu_jl)
    @ccall $sym(get_ptr(ddf)::Ptr{Cfloat}, get_ptr(u)::Ptr{Cfloat}, dx::Cfloat, d::Cint, L::Cint, N::Cint,
                index_mode::Cuint, blocks::Cuint, threads::Cuint)::Cvoid
end

# the C function is float(float*, float*, unsigned); main.jl:28-32 declares it with Float64 by mistake
function inner_product_gpu(v::ROCArray{Float32}, w::ROCArray{Float32}, N)
    sym = Libdl.dlsym(lib, :inner_product_gpu)
    @ccall $sym(get_ptr(v)::Ptr{Cfloat}, get_ptr(w)::Ptr{Cfloat}, N::Cuint)::Cfloat
end

function neighbour_index_gpu(ind, direction, amount, L, d, N, index_mode)::Cint
    sym = Libdl.dlsym(lib, :neighbour_index_gpu)
    @ccall $sym(ind::Cint, direction::Cint, amount::Cint, L::Cint, d::Cint, N::Cint, index_mode::Cint)::Cint
end

function conjugate_gradient_gpu(b::ROCArray{Float32}, x::ROCArray{Float32}, L, d)::Cfloat
    sym = Libdl.dlsym(lib, :conjugate_gradient_gpu)
    @ccall $sym(get_ptr(b)::Ptr{Cfloat}, get_ptr(x)::Ptr{Cfloat}, L::Cint, d::Cint)::Cfloat
end

# additions: layout choice (index_mode 2 = Morton order), cap and tolerance
function conjugate_gradient_gpu_ex(b::ROCArray{Float32}, x::ROCArray{Float32}, L, d; index_mode=0, maxit=200, rel=1e-6)::Cfloat
    sym = Libdl.dlsym(lib, :conjugate_gradient_gpu_ex)
    @ccall $sym(get_ptr(b)::Ptr{Cfloat}, get_ptr(x)::Ptr{Cfloat}, L::Cint, d::Cint, index_mode::Cint, maxit::Cint,
                rel::Cdouble, 1::Cint, 0::Cint, C_NULL::Ptr{Cint}, C_NULL::Ptr{Cfloat}, 0::Cint)::Cfloat
end

function strong_scaling(nblocks, threads_per_block, N, L, d)::Cfloat
    @assert L^d == N
    sym = Libdl.dlsym(lib, :strong_scaling)
    @ccall $sym(nblocks::Cint, threads_per_block::Cint, N::Cint, L::Cint, d::Cint)::Cfloat
end

function time_cpu_laplace(N, L, d)::Cfloat
    @assert L^d == N
    sym = Libdl.dlsym(lib, :time_cpu_laplace)
    @ccall $sym(N::Cint, L::Cint, d::Cint)::Cfloat
end

@testset "indexing on GPU" begin
    @test neighbour_index_gpu(2, 1, 1, 3, 2, 9, 0) == 5
    @test neighbour_index_gpu(2, 0, 1, 3, 2, 9, 0) == 9
    @test neighbour_index_gpu(3, 0, -1, 3, 2, 9, 0) == 9
end;

# 4-D 128^4 in Morton order through the same ccall surface
function morton_demo()
    L, d = 128, 4
    b = AMDGPU.rand(Float32, L^d)
    x = AMDGPU.zeros(Float32, L^d)
    conjugate_gradient_gpu_ex(b, x, L, d; index_mode=2, maxit=200)
end
