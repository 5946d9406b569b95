function [XNK,XLK,PK] = particleSmoother(dynModel,measModel,dynResNorm,odometry,y,...
    x0_nonLin,x0_lin,P0_lin,Q,R,N_P,N_K,dt,sparseFeatures,makePlots)
% PARTICLESMOOTHER - drop-in for the reference src/particleSmoother.m (covariance-form ancestor weights) backed by
% the MI355X HIP library.  Random numbers are drawn here in the reference's order (particleSmoother.m:132-137,149,
% 241,346): for k>1 slot N_P consumes exactly one rand and no randn.  UNTESTED here: no MATLAB in the build image.
  [XNK,XLK,PK] = rbpf_smoother_common(0, dynModel,measModel,dynResNorm,odometry,y,x0_nonLin,x0_lin,P0_lin,Q,R,N_P,N_K,dt, ...
      nargin >= 14 && ~isempty(sparseFeatures) && sparseFeatures);
  if nargin >= 15 && ~isempty(makePlots)
    for k = 1:N_K, makePlots(XNK(:,:,k), XLK(:,k), k, XNK, XLK, PK); end
  end
end
