// shooting.hpp -- host mirror of the reference's BVP driver `shooting` (shooting.hpp:19-385).
//
// Public API verbatim (constructor, Resize, SetMode x2, InitShooting x2, SetDesiredState x2,
// SolveOCP x4, Move x4, SetPrecision, SetContinuationMinStep, GetParameters x3, GetSolution,
// GetCallNumber, Trace, GetModel; all const as in the reference).  Behind it the residual
// F(z) and the forward-difference Jacobian are evaluated on the GPU through the C-ABI
// (socp_residual_batch / socp_fd_jacobian), and the Newton iteration is the library's own
// MINPACK hybrd/hybrj (include/cminpack.h, socp_solver.h) whose finite-difference stage is one
// batched call instead of n sequential callbacks.
#ifndef SOCP_AMD_SHOOTING_HPP_
#define SOCP_AMD_SHOOTING_HPP_

#include <functional>
#include <iostream>
#include <string>
#include <unordered_map>
#include <vector>

#include "commonType.hpp"
#include "model.hpp"

class shooting
{
public:
    // numMulti >= 1 shooting segments; numThread >= 1: the reference splits the segments of a residual over that many
    // std::threads (shooting.cpp:1142-1157).  For a model with device dynamics every segment of every residual row is one GPU
    // trajectory whatever numThread says; for a class without, numThread segment workers run on the host (a persistent pool)
    shooting(model &model, int numMulti = int(1), int numThread = int(1));
    ~shooting();

    void Resize(int numMulti, int numThread) const;
    void SetMode(int const &mode_tf, std::vector<int> const &mode_Xf) const;
    void SetMode(std::vector<int> const &mode_t, std::vector<std::vector<int> > const &mode_X) const;
    void InitShooting(real const &ti, model::mstate const &Xi, real const &tf, model::mstate const &Xf) const;
    void InitShooting(std::vector<real> const &vt, std::vector<model::mstate> const &vX) const;
    void SetDesiredState(real const &ti, model::mstate const &Xi, real const &tf, model::mstate const &Xf) const;
    void SetDesiredState(std::vector<real> const &vt, std::vector<model::mstate> const &vX) const;

    // 1 = converged; continuationStep <= 0: single Newton solve, else discrete continuation
    int SolveOCP(real const &continuationStep) const;
    int SolveOCP(real const &continuationStep, double const &timeoutMS) const;               // -1 on timeout
    int SolveOCP(real const &continuationStep, real &Rdata, real const &Rgoal) const;        // continuation on a parameter
    int SolveOCP(real const &continuationStep, std::string const &Rdata, real const &Rgoal) const
    {
        // shooting.hpp:120-128: an unknown name prints a message (and, in the reference, falls off
        // the end of a non-void function); here it returns 0 = "improper input"
        std::map<std::string, real>::iterator it = myModel.parameters.find(Rdata);
        if (it != myModel.parameters.end()) return SolveOCP(continuationStep, it->second, Rgoal);
        std::cout << std::endl << std::endl << "Data " << Rdata << " does not exist!" << std::endl << std::endl;
        return 0;
    }

    model::mstate Move(real const &ti, model::mstate const &Xi, real const &tf, int isJac = 0) const;
    void Move(real const &ti, model::mstate const &Xi, real const &tf, model::mstate &Xf, int isJac = 0) const;
    model::mstate Move(real const &tf, int isJac = 0) const;
    void Move(real const &tf, model::mstate &Xf, int isJac = 0) const;

    void SetPrecision(real const &xtol) const;
    void SetContinuationMinStep(real const &step) const;
    real GetParameters(int const &k) const;
    real *GetParameters() const;                                  // caller delete[]s (shooting.cpp:463-470)
    void GetParameters(std::vector<real> &paramVector) const;
    void GetSolution(std::vector<real> &vt, std::vector<model::mstate> &vX) const;
    std::vector<int> GetCallNumber() const;                       // {nfev, njev} of the last solve
    void Trace() const;
    model &GetModel() const;

    // ---- additions (not in the reference) -----------------------------------------------------
    // trajectories integrated on the device since construction (the BASELINE metric's unit)
    long long GetTrajectoryCount() const;
    // integrate only the segments an FD column can change (bit-identical Jacobian); default on
    void SetJacobianDedup(bool on) const;
    // the shooting residual F(z) at an arbitrary unknown vector with the current (desired) boundary data: what the Newton
    // solver is shown (StaticShootingFunction, shooting.cpp:859-874) -- on the GPU, or on the host for a model without
    // device dynamics
    std::vector<real> ResidualAt(std::vector<real> const &param) const;

private:
    model &myModel;
    struct data_struct;
    data_struct *data;

    int SolveShooting() const;
    int SolveShootingContinuation(real const &continuationStep) const;
    int SolveShootingContinuation(real const &continuationStep, real &Rdata, real const &Rgoal) const;
    int SolveShootingFunction(int const &numParam, std::vector<real> &param) const;
    void PushProblemToDevice() const;
    void ComputeTimeLine(std::vector<real> const &param, std::vector<real> &timeLine) const;
    void UpdateSolution() const;

    // residual of a model WITHOUT device dynamics (DeviceModelId() == 0): assembled on the host from the model's own virtuals
    // (ComputeTraj, Initial[H]Function, Final[H]Function, SwitchingTimesFunction), shooting.cpp:918-993, 1511-1576
    void HostShootingFunction(std::vector<real> const &param, std::vector<real> &fvec) const;
    static int StaticHostShootingFunction(void *userdata, int n, const real *param, real *fvec, int iflag);
    // the same for modelOrder == 1: analytic Jacobian from the class's own variational equations, solved with hybrj
    // (shooting.cpp:828-852, 877-915, 996-1130)
    void HostShootingFunctionJacobian(std::vector<real> const &param, std::vector<real> &fjac) const;
    void HostMultipleShootingJacobian(real const &t, model::mstate const &X, model::mstate const &Xp, model::mstate const &Xd,
                                      std::vector<int> const &mode_X, int mode_t, std::vector<real> &rows) const;
    static int StaticHostShootingFunctionJacobian(void *userdata, int n, const real *param, real *fvec, real *fjac, int ldfjac, int iflag);
    // segments of one host evaluation: serial, or numThread contiguous blocks (shooting.cpp:1223-1231) over a persistent pool
    void HostForEachSegment(std::function<void(int)> const &segment) const;
    static int StaticShootingFunction(void *userdata, int n, const real *param, real *fvec, int iflag);
    static int StaticShootingFdJacobian(void *userdata, int n, const real *param, const real *fvec, real epsfcn, real *fjac, int ldfjac);
    static int StaticShootingFunctionJacobian(void *userdata, int n, const real *param, real *fvec, real *fjac, int ldfjac, int iflag);
};

#endif
