function [XNK,XLK,PK] = particleSmootherInformationForm(dynModel,measModel,dynResNorm,odometry,y,...
    x0_nonLin,x0_lin,P0_lin,Q,R,N_P,N_K,dt,sparseFeatures,makePlots)
% PARTICLESMOOTHERINFORMATIONFORM - drop-in for the reference src/particleSmootherInformationForm.m backed by the
% MI355X HIP library.  UNTESTED under MATLAB here: no MATLAB in the build image.
  if nargin >= 14 && ~isempty(sparseFeatures) && sparseFeatures
    disp('This code has only been implemented for dense features'); XNK = []; XLK = []; PK = []; return;   % :77-80
  end
  if nargin < 15, makePlots = []; end
  [XNK,XLK,PK] = rbpf_smoother_common(1, dynModel,measModel,dynResNorm,odometry,y,x0_nonLin,x0_lin,P0_lin,Q,R,N_P,N_K,dt,false,makePlots);
end
