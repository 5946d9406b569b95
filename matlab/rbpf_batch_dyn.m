function xn_new = rbpf_batch_dyn(dynModel, xn, dx, dt, Q)
% RBPF_BATCH_DYN - the generic family's dynModel callback: one call from the MEX gateway per time step.
% xn [nN x n] holds the ANCESTORS' states of slots 1..n in slot order; the handle is applied column by column exactly as
% src/particleFilter.m:104-109 / src/particleSmoother.m:132-137 do, so it consumes MATLAB's global random stream itself.
  xn_new = zeros(size(xn));
  for j = 1:size(xn, 2)
    xn_new(:, j) = dynModel(xn(:, j), dx, dt, Q);
  end
end
