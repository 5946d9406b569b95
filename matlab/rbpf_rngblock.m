function rngblk = rbpf_rngblock(kind, nw, N_P, N_T, N_K, smoother)
% RBPF_RNGBLOCK - the random numbers of one filter / smoother call for the MEX gateway (include/rbpf.h `rbpf_rng`), chosen by the
% session options rng_mode / rng_seed (rbpf_options):
%
%   rng_mode 0 (default)  MATLAB's own rand / randn in the reference's interleaved call order -- particleFilter.m:106-108 (per slot
%                         one rand, then dynModel's randn's), particleSmoother.m:132-137,149,241,346 (for k > 1 slot N_P takes
%                         exactly one rand) -- so rng(s,'twister') reproduces the reference run.  An interpreted double loop:
%                         fine at the reference's sizes (N_P = 100), 2e8 iterations at N_P = 65 536, T = 3000.
%   rng_mode 1            MATLAB's stream, vectorised: U = rand(...), Z = randn(...) in two calls.  Same distribution, NOT the
%                         reference's interleaving (a different but equally valid run for the same seed); still O(N_P T) host memory.
%   rng_mode 2            nothing is drawn on the host: the device Philox4x32-10 generator, keyed by rng_seed and the logical
%                         slot / step / iteration (rbpf_options('rng_mode', 2, 'rng_seed', 1)); the throughput path.
% Generic family (kind 4: handles evaluated by MATLAB, which draw their own randn's): only the resampling rand's are needed.
% UNTESTED under MATLAB here: no MATLAB in the build image.
  o = rbpf_mex('options');
  nT = max(N_T-1,0);
  if o.rng_mode == 2
    rngblk = struct('mode','philox','seed',max(o.rng_seed,1));
    return
  end
  if kind == 4 || o.rng_mode == 1
    U = rand(N_P, nT, N_K);
    rngblk = struct('mode','replay','U',U);
    if kind ~= 4, rngblk.Z = randn(nw, N_P, nT, N_K); end
    if smoother, rngblk.Ufin = rand(N_K,1); end
    return
  end
  U = zeros(N_P, nT, N_K); Z = zeros(nw, N_P, nT, N_K); Ufin = zeros(N_K,1);
  for k = 1:N_K
    for t = 1:nT
      if smoother, nOrd = N_P-1; else, nOrd = N_P; end    % the smoothers treat slot N_P separately (particleSmoother.m:132)
      for i = 1:nOrd, U(i,t,k) = rand; Z(:,i,t,k) = randn(nw,1); end
      if smoother
        U(N_P,t,k) = rand;                               % particleSmoother.m:149 (k == 1) / :241 (k > 1)
        if k == 1, Z(:,N_P,t,k) = randn(nw,1); end
      end
    end
    if smoother, Ufin(k) = rand; end                     % :346
  end
  rngblk = struct('mode','replay','U',U,'Z',Z);
  if smoother, rngblk.Ufin = Ufin; end
end
