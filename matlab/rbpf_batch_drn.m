function eDyn = rbpf_batch_drn(dynResNorm, xnkt, xn, dx, dt, Q)
% RBPF_BATCH_DRN - the generic family's dynResNorm callback (src/particleSmoother.m:178-180): eDyn(:,i) is the row the
% handle returns for particle i, one call from the MEX gateway per time step of a smoother iteration k > 1.
  nw = size(Q, 1);
  eDyn = zeros(nw, size(xn, 2));
  for i = 1:size(xn, 2)
    e = dynResNorm(xnkt, xn(:, i), dx, dt, Q);
    eDyn(:, i) = e(:);
  end
end
