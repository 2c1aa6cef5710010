; ModuleID = 'sdcheck2.hip'
source_filename = "sdcheck2.hip"
target datalayout = "e-p:64:64-p1:64:64-p2:32:32-p3:32:32-p4:64:64-p5:32:32-p6:32:32-p7:160:256:256:32-p8:128:128:128:48-p9:192:256:256:32-i64:64-v16:16-v24:32-v32:32-v48:64-v96:128-v192:256-v256:256-v512:512-v1024:1024-v2048:2048-n32:64-S32-A5-G1-ni:7:8:9"
target triple = "amdgcn-amd-amdhsa"

@__hip_cuid_c3f0585dd923f7e6 = addrspace(1) global i8 0
@llvm.compiler.used = appending addrspace(1) global [1 x ptr] [ptr addrspacecast (ptr addrspace(1) @__hip_cuid_c3f0585dd923f7e6 to ptr)], section "llvm.metadata"

; Function Attrs: mustprogress nofree norecurse nosync nounwind willreturn memory(argmem: readwrite)
define protected amdgpu_kernel void @_Z1kPKdPd(ptr addrspace(1) noundef readonly captures(none) %0, ptr addrspace(1) noundef writeonly captures(none) initializes((0, 104)) %1) local_unnamed_addr #0 {
  %3 = load double, ptr addrspace(1) %0, align 8, !tbaa !6
  %4 = getelementptr inbounds nuw i8, ptr addrspace(1) %0, i64 8
  %5 = load double, ptr addrspace(1) %4, align 8, !tbaa !6
  %6 = getelementptr inbounds nuw i8, ptr addrspace(1) %0, i64 16
  %7 = load double, ptr addrspace(1) %6, align 8, !tbaa !6
  %8 = fptrunc double %3 to float
  %9 = tail call noundef float @llvm.fabs.f32(float %8)
  %10 = fpext float %9 to double
  %11 = fptrunc double %5 to float
  %12 = tail call noundef float @llvm.fabs.f32(float %11)
  %13 = fpext float %12 to double
  %14 = fptrunc double %7 to float
  %15 = tail call noundef float @llvm.fabs.f32(float %14)
  %16 = fpext float %15 to double
  %17 = fcmp uge float %9, 1.000000e+06
  %18 = fcmp olt float %9, 1.000000e+06
  %19 = select i1 %18, float %9, float 1.000000e+06
  %20 = fpext float %19 to double
  %21 = sext i1 %17 to i32
  %22 = fcmp ogt float %19, %12
  %23 = select i1 %22, double %13, double %20
  %24 = select i1 %22, i32 1, i32 %21
  %25 = fcmp ogt double %23, %16
  %26 = select i1 %25, i32 2, i32 %24
  switch i32 %26, label %31 [
    i32 0, label %27
    i32 1, label %29
  ]

27:                                               ; preds = %2
  %28 = fneg double %7
  br label %33

29:                                               ; preds = %2
  %30 = fneg double %7
  br label %33

31:                                               ; preds = %2
  %32 = fneg double %5
  br label %33

33:                                               ; preds = %29, %31, %27
  %34 = phi double [ 0.000000e+00, %27 ], [ %30, %29 ], [ %32, %31 ]
  %35 = phi double [ %28, %27 ], [ 0.000000e+00, %29 ], [ %3, %31 ]
  %36 = phi double [ %5, %27 ], [ %3, %29 ], [ 0.000000e+00, %31 ]
  %37 = sitofp i32 %26 to double
  store double %37, ptr addrspace(1) %1, align 8, !tbaa !6
  %38 = getelementptr inbounds nuw i8, ptr addrspace(1) %1, i64 8
  store double %34, ptr addrspace(1) %38, align 8, !tbaa !6
  %39 = getelementptr inbounds nuw i8, ptr addrspace(1) %1, i64 16
  store double %35, ptr addrspace(1) %39, align 8, !tbaa !6
  %40 = getelementptr inbounds nuw i8, ptr addrspace(1) %1, i64 24
  store double %36, ptr addrspace(1) %40, align 8, !tbaa !6
  %41 = getelementptr inbounds nuw i8, ptr addrspace(1) %1, i64 32
  store double %10, ptr addrspace(1) %41, align 8, !tbaa !6
  %42 = getelementptr inbounds nuw i8, ptr addrspace(1) %1, i64 40
  store double %13, ptr addrspace(1) %42, align 8, !tbaa !6
  %43 = getelementptr inbounds nuw i8, ptr addrspace(1) %1, i64 48
  store double %16, ptr addrspace(1) %43, align 8, !tbaa !6
  %44 = fmul double %34, %34
  %45 = fmul double %35, %35
  %46 = fadd double %44, %45
  %47 = fmul double %36, %36
  %48 = fadd double %46, %47
  %49 = tail call noundef double @llvm.sqrt.f64(double %48)
  %50 = tail call noundef double @llvm.fabs.f64(double %49)
  %51 = fcmp ogt double %50, 0x3EB0C6F7A0B5ED8D
  br i1 %51, label %52, label %57

52:                                               ; preds = %33
  %53 = fdiv double 1.000000e+00, %49
  %54 = fmul double %34, %53
  %55 = fmul double %35, %53
  %56 = fmul double %36, %53
  br label %57

57:                                               ; preds = %33, %52
  %58 = phi double [ %54, %52 ], [ %34, %33 ]
  %59 = phi double [ %55, %52 ], [ %35, %33 ]
  %60 = phi double [ %56, %52 ], [ %36, %33 ]
  %61 = getelementptr inbounds nuw i8, ptr addrspace(1) %1, i64 56
  store double %58, ptr addrspace(1) %61, align 8, !tbaa !6
  %62 = getelementptr inbounds nuw i8, ptr addrspace(1) %1, i64 64
  store double %59, ptr addrspace(1) %62, align 8, !tbaa !6
  %63 = getelementptr inbounds nuw i8, ptr addrspace(1) %1, i64 72
  store double %60, ptr addrspace(1) %63, align 8, !tbaa !6
  %64 = fmul double %7, %59
  %65 = fmul double %5, %60
  %66 = fsub double %64, %65
  %67 = fmul double %3, %60
  %68 = fmul double %7, %58
  %69 = fsub double %67, %68
  %70 = fmul double %5, %58
  %71 = fmul double %3, %59
  %72 = fsub double %70, %71
  %73 = fmul double %66, %66
  %74 = fmul double %69, %69
  %75 = fadd double %73, %74
  %76 = fmul double %72, %72
  %77 = fadd double %76, %75
  %78 = tail call noundef double @llvm.sqrt.f64(double %77)
  %79 = tail call noundef double @llvm.fabs.f64(double %78)
  %80 = fcmp ogt double %79, 0x3EB0C6F7A0B5ED8D
  br i1 %80, label %81, label %86

81:                                               ; preds = %57
  %82 = fdiv double 1.000000e+00, %78
  %83 = fmul double %66, %82
  %84 = fmul double %69, %82
  %85 = fmul double %72, %82
  br label %86

86:                                               ; preds = %57, %81
  %87 = phi double [ %83, %81 ], [ %66, %57 ]
  %88 = phi double [ %84, %81 ], [ %69, %57 ]
  %89 = phi double [ %85, %81 ], [ %72, %57 ]
  %90 = getelementptr inbounds nuw i8, ptr addrspace(1) %1, i64 80
  store double %87, ptr addrspace(1) %90, align 8, !tbaa !6
  %91 = getelementptr inbounds nuw i8, ptr addrspace(1) %1, i64 88
  store double %88, ptr addrspace(1) %91, align 8, !tbaa !6
  %92 = getelementptr inbounds nuw i8, ptr addrspace(1) %1, i64 96
  store double %89, ptr addrspace(1) %92, align 8, !tbaa !6
  ret void
}

; Function Attrs: mustprogress nocallback nofree nosync nounwind speculatable willreturn memory(none)
declare float @llvm.fabs.f32(float) #1

; Function Attrs: mustprogress nocallback nofree nosync nounwind speculatable willreturn memory(none)
declare double @llvm.sqrt.f64(double) #1

; Function Attrs: mustprogress nocallback nofree nosync nounwind speculatable willreturn memory(none)
declare double @llvm.fabs.f64(double) #1

attributes #0 = { mustprogress nofree norecurse nosync nounwind willreturn memory(argmem: readwrite) "amdgpu-agpr-alloc"="0" "amdgpu-flat-work-group-size"="1,1024" "amdgpu-no-cluster-id-x" "amdgpu-no-cluster-id-y" "amdgpu-no-cluster-id-z" "amdgpu-no-completion-action" "amdgpu-no-default-queue" "amdgpu-no-dispatch-id" "amdgpu-no-dispatch-ptr" "amdgpu-no-flat-scratch-init" "amdgpu-no-heap-ptr" "amdgpu-no-hostcall-ptr" "amdgpu-no-implicitarg-ptr" "amdgpu-no-lds-kernel-id" "amdgpu-no-multigrid-sync-arg" "amdgpu-no-queue-ptr" "amdgpu-no-workgroup-id-x" "amdgpu-no-workgroup-id-y" "amdgpu-no-workgroup-id-z" "amdgpu-no-workitem-id-x" "amdgpu-no-workitem-id-y" "amdgpu-no-workitem-id-z" "no-trapping-math"="true" "stack-protector-buffer-size"="8" "target-cpu"="gfx950" "target-features"="+16-bit-insts,+ashr-pk-insts,+atomic-buffer-global-pk-add-f16-insts,+atomic-buffer-pk-add-bf16-inst,+atomic-ds-pk-add-16-insts,+atomic-fadd-rtn-insts,+atomic-flat-pk-add-16-insts,+atomic-fmin-fmax-global-f64,+atomic-global-pk-add-bf16-inst,+bf8-cvt-scale-insts,+bitop3-insts,+ci-insts,+dl-insts,+dot1-insts,+dot10-insts,+dot12-insts,+dot13-insts,+dot2-insts,+dot3-insts,+dot4-insts,+dot5-insts,+dot6-insts,+dot7-insts,+dpp,+f16bf16-to-fp6bf6-cvt-scale-insts,+f32-to-f16bf16-cvt-sr-insts,+fp4-cvt-scale-insts,+fp6bf6-cvt-scale-insts,+fp8-conversion-insts,+fp8-cvt-scale-insts,+fp8-insts,+gfx8-insts,+gfx9-insts,+gfx90a-insts,+gfx940-insts,+gfx950-insts,+mai-insts,+permlane16-swap,+permlane32-swap,+prng-inst,+s-memrealtime,+s-memtime-inst,+wavefrontsize64" "uniform-work-group-size"="true" }
attributes #1 = { mustprogress nocallback nofree nosync nounwind speculatable willreturn memory(none) }

!llvm.module.flags = !{!0, !1, !2, !3}
!llvm.ident = !{!4}
!opencl.ocl.version = !{!5}

!0 = !{i32 1, !"amdhsa_code_object_version", i32 600}
!1 = !{i32 1, !"amdgpu_printf_kind", !"hostcall"}
!2 = !{i32 1, !"wchar_size", i32 4}
!3 = !{i32 8, !"PIC Level", i32 2}
!4 = !{!"AMD clang version 22.0.0git (https://github.com/RadeonOpenCompute/llvm-project roc-7.2.0 26014 7b800a19466229b8479a78de19143dc33c3ab9b5)"}
!5 = !{i32 2, i32 0}
!6 = !{!7, !7, i64 0}
!7 = !{!"double", !8, i64 0}
!8 = !{!"omnipotent char", !9, i64 0}
!9 = !{!"Simple C++ TBAA"}
