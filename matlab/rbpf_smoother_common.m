function [XNK,XLK,PK] = rbpf_smoother_common(info_form, dynModel,measModel,dynResNorm,odometry,y,x0_nonLin,x0_lin,P0_lin,Q,R,N_P,N_K,dt,sparse,makePlots)
% Shared marshalling of the two smoother wrappers: handle recognition (rbpf_recognise), random numbers in the reference's
% order (particleSmoother.m:132-137,149,241,346: for k > 1 slot N_P consumes exactly one rand and no randn), one call of
% the MEX gateway.  makePlots runs after every iteration (particleSmoother.m:360-362) through the library's on_step hook.
% UNTESTED under MATLAB here: no MATLAB in the build image.
  desc = rbpf_recognise(dynModel, measModel, dynResNorm, logical(sparse));
  if desc.kind == 3, desc.nLand = size(y, 2); end
  N_T = size(y,1); nw = size(Q,1);
  rngblk = rbpf_rngblock(desc.kind, nw, N_P, N_T, N_K, true);   % rbpf_options rng_mode: exact (default) / vectorised / device Philox
  [XNK,XLK,PK] = rbpf_mex('smoother', desc, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, N_K, dt, rngblk, info_form, makePlots);
end
