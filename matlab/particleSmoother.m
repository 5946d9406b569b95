function [XNK,XLK,PK] = particleSmoother(dynModel,measModel,dynResNorm,odometry,y,...
    x0_nonLin,x0_lin,P0_lin,Q,R,N_P,N_K,dt,sparseFeatures,makePlots)
% PARTICLESMOOTHER - drop-in for the reference src/particleSmoother.m (covariance-form ancestor weights) backed by the
% MI355X HIP library; the reference's example runners call it unchanged (see particleFilter.m in this directory).
% UNTESTED under MATLAB here: no MATLAB in the build image.
  if nargin < 14 || isempty(sparseFeatures), sparseFeatures = false; end
  if nargin < 15, makePlots = []; end
  [XNK,XLK,PK] = rbpf_smoother_common(0, dynModel,measModel,dynResNorm,odometry,y,x0_nonLin,x0_lin,P0_lin,Q,R,N_P,N_K,dt, ...
      sparseFeatures, makePlots);
end
