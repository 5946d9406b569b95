function [XNK,XLK,PK] = rbpf_smoother_common(info_form, dynModel,measModel,dynResNorm,odometry,y,x0_nonLin,x0_lin,P0_lin,Q,R,N_P,N_K,dt,sparse)
% Shared marshalling of the two smoother wrappers.  UNTESTED here: no MATLAB in the build image.
  f = functions(dynModel);
  desc = f.workspace{1}.rbpf_desc;
  if logical(sparse) ~= (desc.kind == 3)
    error('rbpf:usage', 'sparseFeatures must be true for (and only for) the sparse-visual family');
  end
  desc.use_dyn_res_norm = ~isempty(dynResNorm);
  N_T = size(y,1); nw = size(Q,1);
  U = zeros(N_P, max(N_T-1,0), N_K); Z = zeros(nw, N_P, max(N_T-1,0), N_K); Ufin = zeros(N_K,1);
  for k = 1:N_K
    for t = 1:N_T-1
      for i = 1:N_P-1, U(i,t,k) = rand; Z(:,i,t,k) = randn(nw,1); end
      U(N_P,t,k) = rand;                              % particleSmoother.m:149 (k==1) / :241 (k>1)
      if k == 1, Z(:,N_P,t,k) = randn(nw,1); end
    end
    Ufin(k) = rand;                                   % :346
  end
  rngblk = struct('mode','replay','U',U,'Z',Z,'Ufin',Ufin);
  [XNK,XLK,PK] = rbpf_mex('smoother', desc, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, N_K, dt, rngblk, info_form);
end
