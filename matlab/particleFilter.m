function [traj_max,traj_mean,xl_max,xl_mean,P_max,P_mean,traj_sample_iwmax,xn_traj] = ...
    particleFilter(dynModel,measModel,odometry,y,x0_nonLin,x0_lin,P0_lin,Q,R,N_P,dt,sparseFeatures,makePlots)
% PARTICLEFILTER - drop-in for the reference src/particleFilter.m backed by the MI355X HIP library: same signature, same
% outputs, and the reference's example runners call it UNCHANGED (put this directory on the path before the reference's
% src/).  The handles the examples pass are recognised (rbpf_recognise: proposal from functions(h), verified by
% evaluation) and run as HIP kernels; any other handles run through the generic family (called back from the library every
% step).  makePlots is called after every time step with the reference's nine arguments (particleFilter.m:215-217).
%
% Random numbers.  Recognised families: drawn HERE with MATLAB's own rand / randn in the reference's interleaved order
% (particleFilter.m:106-108: per slot one rand, then the randn's of dynModel), so rng(s,'twister') reproduces the reference
% run.  Generic family: the resampling rand's of a step are drawn before that step's dynModel calls (the handle draws its
% own randn's), i.e. the same distribution but not the reference's exact interleaving.  rbpf_options('rng_mode', 1) draws
% vectorised, rbpf_options('rng_mode', 2, 'rng_seed', s) uses the device generator (rbpf_rngblock.m): no O(N_P*T) host loop.
% UNTESTED under MATLAB here: no MATLAB in the build image (the MEX gateway is executed against a mex.h test double).
  if nargin < 12 || isempty(sparseFeatures), sparseFeatures = false; end     % quirk Q1 of the reference's nargin tests
  if nargin < 13, makePlots = []; end
  desc = rbpf_recognise(dynModel, measModel, [], logical(sparseFeatures));
  if desc.kind == 3, desc.nLand = size(y, 2); end
  N_T = size(y,1); nw = size(Q,1);
  rngblk = rbpf_rngblock(desc.kind, nw, N_P, N_T, 1, false);   % rbpf_options rng_mode: exact (default) / vectorised / device Philox
  [traj_max,traj_mean,xl_max,xl_mean,P_max,P_mean,traj_sample_iwmax,xn_traj] = ...
      rbpf_mex('filter', desc, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, dt, rngblk, makePlots);
end
