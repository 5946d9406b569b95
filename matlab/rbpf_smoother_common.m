function [XNK,XLK,PK] = rbpf_smoother_common(info_form, dynModel,measModel,dynResNorm,odometry,y,x0_nonLin,x0_lin,P0_lin,Q,R,N_P,N_K,dt,sparse,makePlots)
% Shared marshalling of the two smoother wrappers: handle recognition (rbpf_recognise), random numbers in the reference's
% order (particleSmoother.m:132-137,149,241,346: for k > 1 slot N_P consumes exactly one rand and no randn), one call of
% the MEX gateway.  makePlots runs after every iteration (particleSmoother.m:360-362) through the library's on_step hook.
% UNTESTED under MATLAB here: no MATLAB in the build image.
  desc = rbpf_recognise(dynModel, measModel, dynResNorm, logical(sparse));
  if desc.kind == 3, desc.nLand = size(y, 2); end
  N_T = size(y,1); nw = size(Q,1);
  U = zeros(N_P, max(N_T-1,0), N_K); Ufin = zeros(N_K,1);
  if desc.kind == 4
    U(:) = rand(size(U)); Ufin(:) = rand(N_K,1);
    rngblk = struct('mode','replay','U',U,'Ufin',Ufin);
  else
    Z = zeros(nw, N_P, max(N_T-1,0), N_K);
    for k = 1:N_K
      for t = 1:N_T-1
        for i = 1:N_P-1, U(i,t,k) = rand; Z(:,i,t,k) = randn(nw,1); end
        U(N_P,t,k) = rand;                              % particleSmoother.m:149 (k==1) / :241 (k>1)
        if k == 1, Z(:,N_P,t,k) = randn(nw,1); end
      end
      Ufin(k) = rand;                                   % :346
    end
    rngblk = struct('mode','replay','U',U,'Z',Z,'Ufin',Ufin);
  end
  [XNK,XLK,PK] = rbpf_mex('smoother', desc, odometry, y, x0_nonLin, x0_lin, P0_lin, Q, R, N_P, N_K, dt, rngblk, info_form, makePlots);
end
