// model.hpp -- host mirror of the reference's plugin class `model` (model.hpp:16-480).
//
// Public surface kept verbatim (class and member names, argument order, defaults, the public data
// members, the FIXED/FREE/CONTINUOUS enum) so existing user programs and model classes compile
// unchanged.  New here is the device hook: a model that has a hand-written gfx950 twin of its
// dynamics reports it through DeviceModelId()/DeviceParams(); ModelInt(), and everything that
// integrates, then runs on the GPU through the C-ABI (include/socp_hip.h); for those models (all in-tree ones) there is
// no CPU path and no GPU is an error.  A user class WITHOUT a device twin -- only the reference's host virtuals -- still
// works: odeTools::integrate runs the reference's loop over its virtual Model() on the host and shooting assembles the
// residual from its virtuals (one-line warning; CPU speed), see odeTools.cpp / shooting.cpp of this mirror.
#ifndef SOCP_AMD_MODEL_HPP_
#define SOCP_AMD_MODEL_HPP_

#include <fstream>
#include <map>
#include <string>

#include "odeTools.hpp"

struct socp_ctx;   // include/socp_hip.h

class model : public odeTools
{
public:
    typedef odeTools::odeVector mstate;
    typedef odeTools::odeVector mcontrol;

    enum { FIXED, FREE, CONTINUOUS };   // model.hpp:34-38

    model(int const &_stateDim, int _modelOrder = 0, int _stepNbr = 10, std::string _fileTrace = std::string(""));
    virtual ~model();

    virtual int GetDim() const { return dim; }

    // model.hpp:77-79
    virtual mstate ComputeTraj(real const &t0, mstate const &X0, real const &tf, int isTrace, int isJac) { return ModelInt(t0, X0, tf, isTrace, isJac); }

    // default residual blocks and their Jacobian forms (model.hpp:90-328)
    virtual void FinalFunction(real const &tf, mstate const &X_tf, mstate const &Xf, std::vector<int> const &mode_X, std::vector<real> &fvec, int isJac) const;
    virtual void FinalHFunction(real const &tf, mstate const &X_tf, mstate const &Xf, std::vector<int> const &mode_X, std::vector<real> &fvec, int isJac) const;
    virtual void InitialFunction(real const &t0, mstate const &X_t0, mstate const &X0, std::vector<int> const &mode_X, std::vector<real> &fvec, int isJac) const;
    virtual void InitialHFunction(real const &t0, mstate const &X_t0, mstate const &X0, std::vector<int> const &mode_X, std::vector<real> &fvec, int isJac) const;
    virtual mstate SwitchingTimesFunction(real const &t, mstate const &X, mstate const &Xp, int isJac) const;
    virtual void SwitchingStateFunction(real const &t, int const &stateID, mstate const &X, mstate const &Xp, mstate const &Xd, mstate &fvec, int isJac) const {}
    virtual void SwitchingTimesUpdate(std::vector<real> const &switchingTimes) {}
    virtual void SetODEIntPrecision(real const &xtol) { odeIntTol = xtol; }

    // ---- public data, as in the reference (model.hpp:359-367; its "protected:" is commented out)
    int dim;
    int modelOrder;
    std::map<std::string, real> parameters;
    std::string strFileTrace;
    int stepNbr;

    virtual mcontrol Control(real const &t, mstate const &X) const = 0;
    virtual mstate Hamiltonian(real const &t, mstate const &X, int isJac) const = 0;
    virtual mstate ModelInt(real const &t0, mstate const &X, real const &tf, int isTrace, int isJac);
    virtual void Trace(real const &t, mstate const &X, std::ofstream &file) const;
    virtual void Trace(real const &t, mstate const &X, std::stringstream &file) const;
    virtual int GetMode(real const &t, mstate const &X) const { return 0; }

    // ---- device hook (new) -------------------------------------------------------------------
    // SOCP_MODEL_* of include/socp_hip.h, or 0 when the class has no device dynamics
    virtual int DeviceModelId() const { return 0; }
    // packed parameter block in the order the device twin expects; returns the count
    virtual int DeviceParams(double *out, int cap) const { (void)out; (void)cap; return 0; }
    // number of RK4 steps per segment the device integrates with (covid19 keeps its own, covid19.cpp:36)
    virtual int DeviceStepNumber() const { return stepNbr; }
    // switching times the control law reads (goddard); empty otherwise
    virtual std::vector<real> DeviceSwitchingTimes() const { return std::vector<real>(); }
    // arithmetic flavour of this model's device kernels: SOCP_VARIANT_AUTO (default: the reference operation order,
    // bit-identical to the CPU path), SOCP_VARIANT_LANE_EXACT, or SOCP_VARIANT_LANE_FAST (restructured arithmetic, within
    // north_star's 1e-8 of it, ~3.6x the throughput -- bench.py reports both).  Without a call the environment variable
    // SOCP_VARIANT=exact|fast decides.  Takes effect at the next solve / evaluation.
    void SetDeviceVariant(int variant) { deviceVariant_ = variant; }
    int GetDeviceVariant() const { return deviceVariant_; }
    // lazily created device context with parameters, step number and switching times refreshed
    socp_ctx *DeviceContext() const;
    // evaluate Model / Control / Hamiltonian (SOCP_EVAL_*) of a device model at one point
    mstate DeviceEval(int what, real const &t, mstate const &X, int isJac) const;

private:
    model() {}
    mutable socp_ctx *deviceCtx_ = nullptr;
    int deviceVariant_ = -1;            // -1: not chosen by the program (the context keeps its default / SOCP_VARIANT)
};

#endif
