function o = rbpf_options(varargin)
% RBPF_OPTIONS - session options of the MI355X library behind particleFilter / particleSmoother /
% particleSmootherInformationForm (include/rbpf.h `rbpf_options`).  They change the schedule or the arithmetic of the device
% code, never the interface, so the example scripts keep calling the three functions unchanged:
%
%   rbpf_options('lazy_depth', 3, 'chol_refresh', 32);   % before run_dense3D_magfield / run_dense2D_withHeading
%   o = rbpf_options();                                   % query
%   rbpf_options('reset');                                % all zero again = the defaults
%
%   lazy_depth    C >= 2: rewrite the stored covariances every C-th step only (filter: <= 4, information form: <= 3);
%                 same algebra, results to rounding
%   chol_refresh  K > 1: carry the ancestor-weight Cholesky factors of particleSmootherInformationForm along the lineages
%                 (rank-1 up/down-dates), refactorise every K-th step: same ancestors and draws, ancestor probabilities within
%                 2e-9, outputs within 1e-9 of 1 = chol(Imat_i + ImatAddt) from scratch at every step, the reference's own
%                 arithmetic; 0 (default): automatic -- 32 for the recognised dense families with nLin >= 128, else 1
%   chol_variant  which kernel factorises (0 automatic); same arithmetic
%   storage       1: covariance banks stored in single precision (arithmetic stays double; 2e-5 instead of 1e-9);
%                 2: double precision, lower block triangle only (particleFilter keeps P symmetric: 0.56 x the memory and
%                 traffic, results within 1e-9; dense-mag filter with 256 / 512 / 1024 basis functions, smoothers with 256 / 512;
%                 dense-radio with 128 basis functions: 0.75 x, pays only without lazy_depth);
%                 3: the lower block triangle in single precision (filter with 512 / 1024 basis functions)
%   n_devices     W > 1: particleFilter / particleSmootherInformationForm shard their N_P particles over W GPUs of this
%                 machine inside the library (one thread per GPU, RCCL collectives); N_P must be a multiple of W; all of the
%                 reference's outputs (particleFilter: makePlots must be empty)
%   device_ids    [1 x W] 0-based HIP device of every rank (default 0 .. W-1); a device named twice makes its ranks share
%                 that GPU over a host-staged transport (a one-GPU machine can so exercise the multi-rank loop)
%   info_rebuild  1: particleSmootherInformationForm with carried factors stores NO information matrix -- every refresh rebuilds
%                 them from the initial matrix along the whole ancestral path (choose chol_refresh in the hundreds); with inplace 1
%                 and lazy_depth 3 the state is 3.6 MB per particle at 515 basis functions: N_P = 65 536 on one 288 GB GPU
%   inplace       1 / -1: force / forbid the single covariance bank rewritten in place (0 automatic; the information-form smoother
%                 takes it on request only)
%   fix_p_mean    1: return the accumulated P_mean instead of the reference's overwritten one (particleFilter.m quirk)
%   jitter        override of the Cholesky retry jitter (0: the reference's 1e-3 / 1e-2)
%   rng_mode      0: MATLAB's rand / randn in the reference's interleaved order (seed-exact, interpreted loop); 1: MATLAB's
%                 stream, vectorised draws; 2: the device Philox generator keyed by rng_seed (rbpf_rngblock.m)
%   rng_seed      seed of the device generator (rng_mode 2)
% UNTESTED under MATLAB here (no MATLAB in the build image); the gateway command is exercised by tests/test_gpu_mex_gateway.py.
  if nargin == 0, o = rbpf_mex('options'); return; end
  if nargin == 1 && ischar(varargin{1}) && strcmp(varargin{1}, 'reset'), o = rbpf_mex('options', struct()); return; end
  if nargin == 1 && isstruct(varargin{1}), o = rbpf_mex('options', varargin{1}); return; end
  cur = rbpf_mex('options');
  for q = 1:2:numel(varargin), cur.(varargin{q}) = varargin{q+1}; end
  o = rbpf_mex('options', cur);
end
