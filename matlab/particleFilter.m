function [traj_max,traj_mean,xl_max,xl_mean,P_max,P_mean,traj_sample_iwmax,xn_traj] = ...
    particleFilter(dynModel,measModel,odometry,y,x0_nonLin,x0_lin,P0_lin,Q,R,N_P,dt,sparseFeatures,makePlots)
% PARTICLEFILTER - drop-in for the reference src/particleFilter.m backed by the MI355X HIP library.
% Same signature and outputs.  dynModel / measModel must be handles made by rbpf_model (a HIP kernel cannot call
% a MATLAB closure).  Seed-exact mode: the random numbers are drawn HERE with MATLAB's own rand / randn in the
% reference's interleaved order (particleFilter.m:106-108: per slot one rand, then the randn's of dynModel), so
% rng(s,'twister') reproduces the reference run.  UNTESTED here: no MATLAB in the build image.
  if nargin < 12 || isempty(sparseFeatures), sparseFeatures = false; end
  if nargin < 13, makePlots = []; end %#ok<NASGU>
  desc = rbpf_descriptor(dynModel, measModel);
  if logical(sparseFeatures) ~= (desc.kind == 3)
    error('rbpf:usage', 'sparseFeatures must be true for (and only for) the sparse-visual family');
  end
  N_T = size(y,1); nw = size(Q,1);
  U = zeros(N_P, max(N_T-1,0)); Z = zeros(nw, N_P, max(N_T-1,0));
  for t = 1:N_T-1
    for i = 1:N_P
      U(i,t) = rand; Z(:,i,t) = randn(nw,1);
    end
  end
  rngblk = struct('mode','replay','U',U,'Z',Z);
  [traj_max,traj_mean,xl_max,xl_mean,P_max,P_mean,traj_sample_iwmax,xn_traj] = ...
      rbpf_mex('filter', desc, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, dt, rngblk);
end

function desc = rbpf_descriptor(dynModel, measModel)
  f = functions(dynModel); g = functions(measModel);
  if ~isfield(f,'workspace') || isempty(f.workspace) || ~isfield(f.workspace{1},'rbpf_desc') || ...
     ~isfield(g,'workspace') || ~isfield(g.workspace{1},'rbpf_desc')
    error('rbpf:unsupported', 'dynModel/measModel must come from rbpf_model (see INTEGRATION.md)');
  end
  desc = f.workspace{1}.rbpf_desc;
end
